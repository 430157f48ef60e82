#!/bin/bash
# rocprofv3 kernel-trace + stats of the headline bench (short run). TAG names the output dir; ENVV = extra env.
# usage: TAG=default [BENCH_ARGS=...] bash tools/gpu_r2_prof.sh
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG:-r2}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps ${STEPS:-3} --warmup 1 --cpu-sample 0 --compare-steps 0 ${BENCH_ARGS} > $OUT/bench.json 2> $OUT/bench.err
cd $GRAFT_REPO_ROOT
find $OUT/raw -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/raw
python - <<'PY'
import csv, os
p = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "prof_" + os.environ.get("TAG", "r2"), "kernel_stats.csv")
rows = list(csv.DictReader(open(p)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time ms", tot/1e6)
for r in rows[:22]:
    print("%-100s calls=%6s total_ms=%9.2f avg_us=%10.1f pct=%5.1f" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
