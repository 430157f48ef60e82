#!/bin/bash
# Round 6: kernel trace of the NODATA workload (10000^2, 15 %, mask 2468, tau 0.06) with the enriched levels on the fused pass
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6s
rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "enriched_levels" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 600 python tools/nodata_iters.py 10000 2468,1 0.06 > $OUT/nodata_10000.jsonl 2> $OUT/nd.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/tools/nodata_iters.py 10000 2468 0.06 > $OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, os, json
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r6s")
for ln in open(out + "/nodata_10000.jsonl"):
    d = json.loads(ln); print("seed %5d tau %.2f iters %.2f/%d ms16 %.1f setup %.0f" % (d["mask_seed"], d["tau"], d["iters_mean"], d["iters_max"], d["ms_per_16_pairs"], d["setup_device_ms"]))
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:40]:
        print(r["Name"][:90], r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 1), round(float(r["AverageNs"]) / 1e3, 1))
    os.system("cp %s %s/kernel_stats_nodata.csv" % (f, out))
PY
find $OUT -name "*.csv" -size +2M -delete
