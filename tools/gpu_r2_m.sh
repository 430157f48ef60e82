#!/bin/bash
# Round 2, call M: sweeps per coarse level (level 1 vs deeper levels) -- iteration counts and time per batch.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2m
rm -rf $OUT; mkdir -p $OUT
B="python bench.py --compare-steps 0 --cpu-sample 0 --steps 4"
for cfg in "2 2" "1 2" "1 3" "1 4" "2 3" "1 6"; do
  set -- $cfg
  CSGPU_NU_L1=$1 CSGPU_NU_DEEP=$2 timeout 200 $B > $OUT/l1_$1_deep_$2.json 2> $OUT/l1_$1_deep_$2.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2m/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "ms/step %.1f iters %.2f max %d relres %.2e" % (d["ms_per_step"], d["iters_mean"], d["iters_max"], d["max_relres"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e, open(f[:-5]+".err").read()[-300:])
PY
