#!/bin/bash
# Round 6, fourth GPU call: what a locality-restoring renumbering of a network would buy (host RCM / true coordinates against
# the ids as given), and rocprofv3 PMC traffic of the CSR SpMM on the geometric network.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6d
rm -rf $OUT; mkdir -p $OUT
timeout 900 python tools/network_renumber_ab.py 1000000 > $OUT/renumber_1e6.jsonl 2> $OUT/renumber_1e6.err; cut -c1-900 $OUT/renumber_1e6.jsonl
timeout 1200 python tools/network_renumber_ab.py 5000000 > $OUT/renumber_5e6.jsonl 2> $OUT/renumber_5e6.err; cut -c1-900 $OUT/renumber_5e6.jsonl
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/geo_pmc_$c -o p -- python $GRAFT_REPO_ROOT/tools/network_renumber_ab.py 1000000 > $OUT/geo_pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, os, collections, json
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r6d")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/geo_pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "spmv_kernel<double, 16" in row.get("Kernel_Name", ""):
            agg[row["Kernel_Name"][:100]][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {c: {"n": len(v), "values_sorted_head": sorted(v)[:3], "values_sorted_tail": sorted(v)[-3:]} for c, v in d.items()} for k, d in agg.items()}
json.dump(res, open(out + "/geo_pmc_spmv_k16.json", "w"), indent=1)
print(json.dumps(res)[:3000])
PY
find $OUT -name "*.csv" -size +4M -delete
