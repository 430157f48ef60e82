#!/bin/bash
# PMC passes over the headline bench (short): FETCH_SIZE / WRITE_SIZE per kernel, separate passes, kernel-trace only.
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_bench
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-sample 0 --compare-steps 0 ${BENCH_ARGS} > $OUT/p$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, os, json
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_bench")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row.get("Kernel_Name", "")[:100]][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
for k, d in agg.items():
    # mean_fullsize: launches of the full-size workload only (the bench's 768^2 warm-up solves launch the same kernels)
    res[k] = {c: {"n": len(v), "mean": sum(v) / len(v), "max": max(v),
                  "n_fullsize": len([x for x in v if x >= 0.5 * max(v)]),
                  "mean_fullsize": (lambda w: sum(w) / len(w))([x for x in v if x >= 0.5 * max(v)])} for c, v in d.items()}
json.dump(res, open(out + "/pmc_by_kernel.json", "w"), indent=1)
for k, d in sorted(res.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", {"max": 0})["max"])[:14]:
    print(k[:95], {c: (v["n"], round(v["max"])) for c, v in d.items()})
PY
find $OUT -name "*.csv" -size +5M -delete
