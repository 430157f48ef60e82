#!/bin/bash
# Coarsest-level pseudo-inverse of fp32 hierarchies: is the near-null mode kept (noise eigenvalue above the cutoff)?
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2pinv
rm -rf $OUT; mkdir -p $OUT
for cut in default 1e-4; do
  if [ $cut = default ]; then unset CSGPU_PINV_CUT; else export CSGPU_PINV_CUT=$cut; fi
  CSGPU_VERBOSE=1 timeout 600 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --compare-steps 5 --host-csr 0 > $OUT/b_$cut.json 2> $OUT/b_$cut.err
  grep "coarsest level" $OUT/b_$cut.err | sort | uniq -c | head -8
  python - $OUT/b_$cut.json $cut <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("cut", sys.argv[2], {k: d.get(k) for k in ("value", "ms_per_step", "iters_mean", "iters_max", "max_relres")}, "fp64:", {k: d["fp64_path"].get(k) for k in ("ms_per_step", "iters_mean")})
PY
  timeout 300 python tools/tail_probe.py 2>&1 | grep '"pb": 4, "graph": 0' | cut -c1-330
done
