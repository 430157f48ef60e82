#!/bin/bash
# Round 5: the fused first-two-sweeps pass after its prefetch fix (scaling moved from the load to the LDS store): level-1
# launches at 6000^2 (rocprofv3) and the NODATA timings at 6000^2 / 10000^2, against CSGPU_DIA25_NO_J0=1.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5m
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PAIRS=32 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o nd -- python $GRAFT_REPO_ROOT/tools/nodata_iters.py 6000 2468 0.06 > $OUT/nd6000.jsonl 2> $OUT/err.log
find $OUT/raw -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_6000.csv \;
rm -rf $OUT/raw
cd $GRAFT_REPO_ROOT
for j in 1 0; do
  if [ $j = 1 ]; then export CSGPU_DIA25_NO_J0=1; else unset CSGPU_DIA25_NO_J0; fi
  timeout 300 python tools/nodata_iters.py 6000 2468 0.06 >> $OUT/ab_$j.jsonl 2>> $OUT/err.log
  timeout 300 python tools/nodata_iters.py 10000 2468 0.06 >> $OUT/ab_$j.jsonl 2>> $OUT/err.log
done
python - <<'PY'
import csv, json, os
o = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r5m/"
for r in csv.DictReader(open(o + "kernel_stats_6000.csv")):
    if "dia25w" in r["Name"] or "scale_dinv" in r["Name"]:
        print("   ", r["Name"][12:62], "calls", r["Calls"], "avg %.3f ms max %.3f ms" % (float(r["AverageNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
for j in (1, 0):
    for ln in open(o + "ab_%d.jsonl" % j):
        d = json.loads(ln); print("NO_J0=%d N %d iters %.2f ms16 %.1f R0 %.15g" % (j, d["N"], d["iters_mean"], d["ms_per_16_pairs"], d["R0"]))
PY
