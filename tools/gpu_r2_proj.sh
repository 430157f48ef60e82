#!/bin/bash
# Candidate projected out of the coarse tail's right-hand sides: does it let the fp32 hierarchy of a 10000^2 raster use
# the Chebyshev weights?
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2proj
rm -rf $OUT; mkdir -p $OUT
for tag in jacobi_proj cheb_proj; do
  case $tag in jacobi_proj) unset CSGPU_COARSE_CHEBYSHEV;; cheb_proj) export CSGPU_COARSE_CHEBYSHEV=1;; esac
  timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --compare-steps 0 --host-csr 0 > $OUT/$tag.json 2> $OUT/$tag.err
  python - $OUT/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], {k: d.get(k) for k in ("ms_per_step", "iters_mean", "iters_max", "max_relres")})
PY
done
unset CSGPU_COARSE_CHEBYSHEV
timeout 200 python -m pytest tests -m gpu -q -x -k "coarse_tail or near_kernel or grounded or chebyshev or onetoall" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
