#!/bin/bash
# rocprofv3 kernel-trace + stats of the headline bench (short run). TAG names the output dir; ENVV = extra env.
# usage: TAG=default [BENCH_ARGS=...] bash tools/gpu_r2_prof.sh
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG:-r2}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps ${STEPS:-3} --warmup 1 --cpu-sample 0 --compare-steps 0 ${BENCH_ARGS} > $OUT/bench.json 2> $OUT/bench.err
cd $GRAFT_REPO_ROOT
find $OUT/raw -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/raw -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
# The run also holds bench.py's untimed 768^2 warm-up solves (same kernels, microsecond launches), which drag the
# per-kernel averages of kernel_stats.csv down: fullsize.json = the same statistics over the launches of the 10000^2
# workload only (grid size of the largest launch of each kernel), the figure roofline.avg_ms has to agree with.
python - $OUT/kernel_trace.csv $OUT/kernel_stats_fullsize.json <<'PY'
import csv, json, sys, collections
by = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    grid = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)
    by[r["Kernel_Name"]].append((grid, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
out = []
for name, ls in by.items():
    gmax = max(g for g, _ in ls)
    full = [d for g, d in ls if g == gmax]
    real = [d for d in full if d > 0.25 * max(full)]      # launches enqueued after convergence exit on a device flag
    out.append({"kernel": name[:140], "calls_all": len(ls), "calls_fullsize": len(full), "calls_fullsize_real": len(real),
                "avg_ms_all": sum(d for _, d in ls) / len(ls) / 1e6, "avg_ms_fullsize_real": sum(real) / len(real) / 1e6,
                "total_ms_all": sum(d for _, d in ls) / 1e6})
out.sort(key=lambda e: -e["total_ms_all"])
json.dump(out[:30], open(sys.argv[2], "w"), indent=1)
for e in out[:8]:
    print("%-90s all=%4d avg %.3f ms | full-size real=%4d avg %.3f ms" % (e["kernel"][:90], e["calls_all"], e["avg_ms_all"], e["calls_fullsize_real"], e["avg_ms_fullsize_real"]))
PY
rm -rf $OUT/raw $OUT/kernel_trace.csv
python - <<'PY'
import csv, os
p = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "prof_" + os.environ.get("TAG", "r2"), "kernel_stats.csv")
rows = list(csv.DictReader(open(p)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time ms", tot/1e6)
for r in rows[:22]:
    print("%-100s calls=%6s total_ms=%9.2f avg_us=%10.1f pct=%5.1f" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
