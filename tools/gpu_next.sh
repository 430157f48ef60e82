#!/bin/bash
# First GPU session of the next round: time the experiments prepared at the end of round 1 against the default build.
# (~3 GPU-minutes.) Outputs -> gpurun_out/next/
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/next
rm -rf $OUT; mkdir -p $OUT
timeout 400 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
B="python bench.py --compare-steps 0 --cpu-sample 0 --steps 5"
for k in 16 8; do
  timeout 200 $B --batch $k > $OUT/default_k$k.json 2> $OUT/default_k$k.err
  CSGPU_WAVE_SPMM=1 timeout 200 $B --batch $k > $OUT/wave_k$k.json 2> $OUT/wave_k$k.err
done
# the wave variant through the parity tests that exercise the two products
CSGPU_WAVE_SPMM=1 timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "level_products or pairs_match or graph_replay" > $OUT/pytest_wave.log 2>&1; tail -1 $OUT/pytest_wave.log
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/next/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "value %.2f ms/step %.1f cg_spmm_ms %.3f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_ms"], d["roofline"]["frac"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
