#!/bin/bash
# Round 5, third GPU call: the enrichment passes from precomputed sparse forms -- cost and iterations, fp64 and mixed.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5c
rm -rf $OUT; mkdir -p $OUT
timeout 600 python tools/nodata_iters.py 3000 2468,1 0,0.06,0.1 > $OUT/nodata_3000.jsonl 2> $OUT/err3000.log
timeout 1200 python tools/nodata_iters.py 10000 2468,1,2 0,0.04,0.06,0.08,0.1 > $OUT/nodata_10000.jsonl 2> $OUT/err10000.log
PB=4 timeout 600 python tools/nodata_iters.py 10000 2468,1 0,0.06,0.1 > $OUT/nodata_10000_mixed.jsonl 2> $OUT/err10000m.log
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5c/*.jsonl")):
    print(os.path.basename(f))
    for ln in open(f):
        d=json.loads(ln); print("  seed %5d tau %.2f iters %.2f/%d ms16 %.1f setup %.0f ms nc %d" % (d["mask_seed"],d["tau"],d["iters_mean"],d["iters_max"],d["ms_per_16_pairs"],d["setup_device_ms"],d["not_converged"]))
PY
tail -n 3 $OUT/err*.log
