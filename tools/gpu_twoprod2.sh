#!/bin/bash
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/twoprod2
rm -rf $OUT; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "level_products or two_product or pairs_match" > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
if ! grep -q " passed" $OUT/pytest_gpu.log || grep -q "failed" $OUT/pytest_gpu.log; then echo "GPU TESTS FAILED"; exit 1; fi
B="python bench.py --compare-steps 0 --cpu-sample 0 --steps 4 --opt itmax=200"
timeout 300 $B --batch 16 > $OUT/b16_on.json 2> $OUT/b16_on.err
CSGPU_NO_LONGROW=1 timeout 300 $B --batch 16 > $OUT/b16_nolong.json 2> $OUT/b16_nolong.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/twoprod2/b*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "value %.2f ms/step %.1f" % (d["value"], d["ms_per_step"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
BENCH_ARGS="--compare-steps 0 --opt itmax=200" STEPS=2 bash tools/gpu_prof.sh > $OUT/prof_on.txt 2>&1; head -24 $OUT/prof_on.txt | cut -c1-220
cp gpurun_out/prof/kernel_stats.csv $OUT/kernel_stats_on.csv
