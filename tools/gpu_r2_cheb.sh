#!/bin/bash
# Chebyshev weights on the coarse levels (default) vs damped Jacobi there (CSGPU_COARSE_JACOBI=1): GPU tests, the bench
# raster at three sizes, and the heterogeneous probe (tools/hetero_probe.py).
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2cheb
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
for s in 1000 3000 5000 10000; do for d in cheb jacobi; do
  if [ $d = jacobi ]; then export CSGPU_COARSE_JACOBI=1; else unset CSGPU_COARSE_JACOBI; fi
  st=20; [ $s -ge 3000 ] && st=6
  timeout 600 python bench.py --size $s --steps $st --warmup 2 --cpu-sample 0 --compare-steps 0 --host-csr 0 > $OUT/s${s}_$d.json 2> $OUT/s${s}_$d.err
  python - $OUT/s${s}_$d.json $s $d <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("size", sys.argv[2], sys.argv[3], {k: d.get(k) for k in ("value", "ms_per_step", "iters_mean", "iters_max", "max_relres", "setup_device_s")})
PY
done; done
for d in cheb jacobi; do
  if [ $d = jacobi ]; then export CSGPU_COARSE_JACOBI=1; else unset CSGPU_COARSE_JACOBI; fi
  timeout 600 python tools/hetero_probe.py 3000 > $OUT/hetero_$d.jsonl 2> $OUT/hetero_$d.err; cat $OUT/hetero_$d.jsonl | cut -c1-300
done
