#!/bin/bash
# Round 6: enrichment threshold with the enriched levels on the fused pass, 5 mask seeds at 10000^2 (tau 0 = no enrichment)
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6r
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python tools/nodata_iters.py 10000 2468,1,2,3,4 0,0.06,0.08,0.10 > $OUT/nodata_10000_tau_fused.jsonl 2> $OUT/nd.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6r/nodata*.jsonl")):
    print(os.path.basename(f))
    for ln in open(f):
        d=json.loads(ln); print("  seed %5d tau %.2f iters %.2f/%d ms16 %.1f setup %.0f ms nc %d fused %d" % (d["mask_seed"],d["tau"],d["iters_mean"],d["iters_max"],d["ms_per_16_pairs"],d["setup_device_ms"],d["not_converged"],d["fused_restrict_solves"]))
PY
tail -3 $OUT/nd.err
