#!/bin/bash
# Round 6, evidence run of the closing build after the enriched levels joined the fused pass: whole device suite, smoke, the driver's bench command (live PMC traffic, an
# oracle figure on every leg), rocprofv3 kernel stats of the fp64 path at full size, FETCH_SIZE / WRITE_SIZE passes of both
# paths (+ pmc_update.py), NODATA iteration counts over 5 mask seeds at 3000^2 and 10000^2 with the enrichment off / on.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6final5
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest_gpu.log 2>&1; tail -14 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 2500 $OUT/bench.json
TAG=r6final5_fp64 STEPS=3 BENCH_ARGS="--precond same --host-csr 0 --extra-legs 0 --pmc-live 0" bash tools/gpu_r2_prof.sh > $OUT/prof_fp64.log 2>&1; head -10 $OUT/prof_fp64.log
cp gpurun_out/prof_r6final5_fp64/kernel_stats.csv $OUT/kernel_stats_fp64.csv
cp gpurun_out/prof_r6final5_fp64/kernel_stats_fullsize.json $OUT/kernel_stats_fp64_fullsize.json
BENCH_ARGS="--precond same --host-csr 0 --extra-legs 0 --pmc-live 0" bash tools/gpu_pmc.sh > $OUT/pmc_fp64.log 2>&1; tail -6 $OUT/pmc_fp64.log
cp gpurun_out/pmc_bench/pmc_by_kernel.json $OUT/pmc_by_kernel_fp64.json
BENCH_ARGS="--precond fp32 --host-csr 0 --extra-legs 0 --pmc-live 0" bash tools/gpu_pmc.sh > $OUT/pmc_mixed.log 2>&1; tail -6 $OUT/pmc_mixed.log
cp gpurun_out/pmc_bench/pmc_by_kernel.json $OUT/pmc_by_kernel_mixed.json
timeout 400 python tools/nodata_iters.py 3000 2468,1,2,3,4 0,0.06 > $OUT/nodata_3000_5seeds.jsonl 2> $OUT/nd.err
timeout 900 python tools/nodata_iters.py 10000 2468,1,2,3,4 0,0.06 > $OUT/nodata_10000_5seeds.jsonl 2>> $OUT/nd.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6final5/nodata*.jsonl")):
    print(os.path.basename(f))
    for ln in open(f):
        d=json.loads(ln); print("  seed %5d tau %.2f iters %.2f/%d ms16 %.1f setup %.0f ms nc %d" % (d["mask_seed"],d["tau"],d["iters_mean"],d["iters_max"],d["ms_per_16_pairs"],d["setup_device_ms"],d["not_converged"]))
PY
# BASELINE configs[4] as its own workload (closing build after the enriched levels joined the fused pass) + rocprofv3 kernel stats of it
timeout 600 python bench.py --workload network --gpus 1 --steps 3 --warmup 1 > $OUT/bench_network.json 2> $OUT/bench_network.err; tail -c 700 $OUT/bench_network.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/net_trace -o net -- python $GRAFT_REPO_ROOT/bench.py --workload network --steps 2 --warmup 1 > $OUT/net_trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/net_pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --workload network --steps 1 --warmup 1 > $OUT/net_pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, os, collections, json
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r6final5")
for f in glob.glob(out + "/net_trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:10]:
        print(r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/net_pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "spmv_kernel" in row.get("Kernel_Name", ""):
            agg[row["Kernel_Name"][:110]][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {c: {"n": len(v), "mean": sum(v) / len(v), "max": max(v)} for c, v in d.items()} for k, d in agg.items()}
json.dump(res, open(out + "/net_pmc_by_kernel.json", "w"), indent=1)
for k, d in sorted(res.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", {"max": 0})["max"])[:6]:
    print(k[:95], {c: (v["n"], round(v["mean"]), round(v["max"])) for c, v in d.items()})
PY
export CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu.so
timeout 600 python tools/fuzz_enrich_fused.py 77 100 > $OUT/fuzz_enrich_fused_77.jsonl 2> $OUT/f77.err; tail -1 $OUT/fuzz_enrich_fused_77.jsonl
unset CSGPU_LIB
find $OUT -name "*.csv" -size +4M -delete
