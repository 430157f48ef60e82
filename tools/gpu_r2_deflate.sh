#!/bin/bash
# fp32 hierarchies: near-kernel candidate projected out of the coarsest operator (default) vs kept (CSGPU_NO_DEFLATION=1)
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2deflate
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x -k "coarse_tail or near_kernel" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
for s in 300 1000 2000 5000 10000; do for d in on off; do
  if [ $d = off ]; then export CSGPU_NO_DEFLATION=1; else unset CSGPU_NO_DEFLATION; fi
  st=20; [ $s -ge 5000 ] && st=6
  timeout 600 python bench.py --size $s --steps $st --warmup 2 --cpu-sample 0 --compare-steps $st --host-csr 0 > $OUT/s${s}_$d.json 2> $OUT/s${s}_$d.err
  python - $OUT/s${s}_$d.json $s $d <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("size", sys.argv[2], "deflation", sys.argv[3], {k: d.get(k) for k in ("value", "ms_per_step", "iters_mean", "iters_max", "max_relres")}, "fp64:", {k: d["fp64_path"].get(k) for k in ("ms_per_step", "iters_mean", "max_rel_diff_R_vs_mixed_path")})
PY
done; done
unset CSGPU_NO_DEFLATION
timeout 300 python tools/tail_probe.py 2>&1 | grep '"pb": 4, "graph": 0' | cut -c1-330
