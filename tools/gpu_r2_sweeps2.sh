#!/bin/bash
# Jacobi sweeps per coarse level, re-tuned after the near-kernel fix (10000^2).
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2sweeps2
rm -rf $OUT; mkdir -p $OUT
for cfg in "2 3" "1 3" "1 2" "2 2" "3 3" "1 1" "2 3"; do
  set -- $cfg
  tag=l1_$1_deep_$2; [ -f $OUT/$tag.json ] && tag=${tag}_b
  CSGPU_NU_L1=$1 CSGPU_NU_DEEP=$2 timeout 600 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --compare-steps 0 --host-csr 0 > $OUT/$tag.json 2> $OUT/$tag.err
  python - $OUT/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], {k: d.get(k) for k in ("value", "ms_per_step", "iters_mean", "iters_max", "max_relres")})
PY
done
