#!/bin/bash
# Round 6, first GPU call: the whole GPU suite, the driver's bench command, BASELINE configs[4] as its own workload
# (bench.py --workload network), rocprofv3 kernel trace + PMC passes of that workload (profiles/r6_network_*).
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6a
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest_gpu.log 2>&1; tail -14 $OUT/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.json
timeout 600 python bench.py --workload network --gpus 1 --steps 3 --warmup 1 > $OUT/bench_network.json 2> $OUT/bench_network.err; tail -c 1500 $OUT/bench_network.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/net_trace -o net -- python $GRAFT_REPO_ROOT/bench.py --workload network --steps 2 --warmup 1 > $OUT/net_trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/net_pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --workload network --steps 1 --warmup 1 > $OUT/net_pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, os, collections, json
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r6a")
for f in glob.glob(out + "/net_trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/net_pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row.get("Kernel_Name", "")[:110]][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {c: {"n": len(v), "mean": sum(v) / len(v), "max": max(v)} for c, v in d.items()} for k, d in agg.items()}
json.dump(res, open(out + "/net_pmc_by_kernel.json", "w"), indent=1)
for k, d in sorted(res.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", {"max": 0})["max"])[:8]:
    print(k[:95], {c: (v["n"], round(v["mean"]), round(v["max"])) for c, v in d.items()})
PY
find $OUT -name "*.csv" -size +4M -delete
