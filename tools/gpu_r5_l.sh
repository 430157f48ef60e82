#!/bin/bash
# Round 5: the 25-point kernel's level-1 launches at 6000^2 and 10000^2 (rocprofv3), and the tile width knob (experimental
# build libcsgpu_seg.so: CSGPU_DIA25_SEG) -- VERDICT r4 item 4's criterion is stated at 6000^2.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5l
rm -rf $OUT; mkdir -p $OUT
export CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu_seg.so
cd /tmp && export TMPDIR=/tmp
for seg in 32 64 128; do
  CSGPU_DIA25_SEG=$seg PAIRS=32 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw$seg -o nd -- python $GRAFT_REPO_ROOT/tools/nodata_iters.py 6000 2468 0.06 > $OUT/nd6000_$seg.jsonl 2> $OUT/err.log
  find $OUT/raw$seg -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_6000_seg$seg.csv \;
  rm -rf $OUT/raw$seg
done
cd $GRAFT_REPO_ROOT
for seg in 32 64 128; do
  CSGPU_DIA25_SEG=$seg PAIRS=64 timeout 300 python tools/nodata_iters.py 10000 2468 0.06 > $OUT/nd10000_$seg.jsonl 2>> $OUT/err.log
done
python - <<'PY'
import csv, json, os
o = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r5l/"
for seg in (32, 64, 128):
    rows = [r for r in csv.DictReader(open(o + "kernel_stats_6000_seg%d.csv" % seg)) if "dia25w" in r["Name"]]
    d6 = json.loads(open(o + "nd6000_%d.jsonl" % seg).read().strip().splitlines()[-1])
    d10 = json.loads(open(o + "nd10000_%d.jsonl" % seg).read().strip().splitlines()[-1])
    print("seg", seg, "6000^2 ms16 %.1f | 10000^2 ms16 %.1f" % (d6["ms_per_16_pairs"], d10["ms_per_16_pairs"]))
    for r in rows:
        print("   ", r["Name"][13:60], "calls", r["Calls"], "avg %.3f ms max %.3f ms" % (float(r["AverageNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
PY
tail -n 3 $OUT/err.log
