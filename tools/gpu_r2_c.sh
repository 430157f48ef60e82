#!/bin/bash
# Round 2, call C: lattice S product of the two-product level vs the CSR [S Q] product; tile-width sweep.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c
rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "level_products or paths_agree or config2 or lattice" > $OUT/pytest_parity.log 2>&1; tail -3 $OUT/pytest_parity.log
B="python bench.py --compare-steps 0 --cpu-sample 0 --steps 5"
for seg in 8 16 24 32 48 64; do
  CSGPU_DIA_SEG=$seg timeout 200 $B > $OUT/seg$seg.json 2> $OUT/seg$seg.err
done
CSGPU_DIA_SEG=32 CSGPU_NO_LATTICE_S=1 timeout 200 $B > $OUT/seg32_csrM.json 2> $OUT/seg32_csrM.err
CSGPU_DIA_SEG=32 timeout 200 $B --precond same > $OUT/seg32_fp64.json 2> $OUT/seg32_fp64.err
cd /tmp && export TMPDIR=/tmp
CSGPU_DIA_SEG=32 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r2c -- python $GRAFT_REPO_ROOT/bench.py --compare-steps 0 --cpu-sample 0 --steps 3 > $OUT/prof_bench.json 2> $OUT/prof_bench.err
cd $GRAFT_REPO_ROOT
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/prof -type f ! -name "*stats*" -delete
python - <<'PY'
import json, glob, os, csv
root=os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2c/"
for f in sorted(glob.glob(root+"*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "value %.2f ms/step %.1f cg_prod_ms %.3f iters %.2f relres %.2e setup %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_ms"], d["iters_mean"], d["max_relres"], d["setup_s"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e, open(f[:-5]+".err").read()[-500:])
try:
    rows=list(csv.DictReader(open(root+"kernel_stats.csv")))
    for r in rows[:16]:
        print("%-90s calls %6s avg_us %10.1f pct %5s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
except Exception as e:
    print("no stats", e)
PY
