#!/bin/bash
# Does the number of workgroups per SpMM launch (how far co-resident workgroups can drift apart) change the over-fetch?
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/gridcap
rm -rf $OUT; mkdir -p $OUT
B="python bench.py --compare-steps 0 --cpu-sample 0 --steps 4 --opt itmax=200"
for cap in 16384 65536 262144; do
CSGPU_SPMV_GRID_CAP=$cap timeout 300 $B --batch 16 > $OUT/cap_$cap.json 2> $OUT/cap_$cap.err
done

python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/gridcap/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "value %.2f ms/step %.1f spmm_ms %.3f solve_only %.2f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_ms"], d["solve_only_pairs_per_s"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
