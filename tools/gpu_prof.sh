#!/bin/bash
# rocprofv3 kernel-trace + stats of the headline bench (short run); summary -> gpurun_out/prof/
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps ${STEPS:-2} --warmup 1 --cpu-sample 0 ${BENCH_ARGS} > $OUT/bench.json 2> $OUT/bench.err
cd $GRAFT_REPO_ROOT
find $OUT/raw -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/raw -name "*kernel_trace.csv" -size +30M -delete
python - <<'PY'
import csv, os
p = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "prof", "kernel_stats.csv")
rows = list(csv.DictReader(open(p)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time ms", tot/1e6)
for r in rows[:40]:
    print("%-110s calls=%6s total_ms=%9.2f avg_us=%10.1f pct=%5.1f" % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
cat $OUT/bench.json | head -c 600
