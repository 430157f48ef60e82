#!/bin/bash
# Round 5, fourth GPU call: rocprofv3 kernel stats of a NODATA call with the enrichment on (which pass costs what), then
# timings off / on for both precisions after the post pass got a full grid.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5d
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CSGPU_VERBOSE=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o nd -- python $GRAFT_REPO_ROOT/tools/nodata_iters.py 10000 2468 0.06 > $OUT/prof.jsonl 2> $OUT/prof.err
cd $GRAFT_REPO_ROOT
find $OUT/raw -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/raw
head -45 $OUT/kernel_stats.csv | cut -c1-200
grep enrichment $OUT/prof.err
timeout 900 python tools/nodata_iters.py 10000 2468,1 0,0.06,0.1 > $OUT/nodata_10000.jsonl 2> $OUT/err10000.log
PB=4 timeout 600 python tools/nodata_iters.py 10000 2468,1 0,0.06 > $OUT/nodata_10000_mixed.jsonl 2> $OUT/err10000m.log
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5d/nodata*.jsonl")):
    print(os.path.basename(f))
    for ln in open(f):
        d=json.loads(ln); print("  seed %5d tau %.2f iters %.2f/%d ms16 %.1f setup %.0f ms nc %d" % (d["mask_seed"],d["tau"],d["iters_mean"],d["iters_max"],d["ms_per_16_pairs"],d["setup_device_ms"],d["not_converged"]))
PY
tail -n 3 $OUT/err*.log
