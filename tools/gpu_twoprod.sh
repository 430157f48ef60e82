#!/bin/bash
# two-product level / long-row kernel A-B. Outputs -> gpurun_out/twoprod/
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/twoprod
rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
if ! grep -q " passed" $OUT/pytest_gpu.log || grep -q "failed" $OUT/pytest_gpu.log; then echo "GPU TESTS FAILED - skipping benches"; exit 1; fi
B="python bench.py --compare-steps 0 --cpu-sample 0 --steps 4 --opt itmax=200"
timeout 300 $B --batch 16 > $OUT/b16_on.json 2> $OUT/b16_on.err
timeout 300 $B --batch 16 --opt two_product=-1 > $OUT/b16_off.json 2> $OUT/b16_off.err
CSGPU_NARROW_TILE=1 timeout 300 $B --batch 16 > $OUT/b16_narrow.json 2> $OUT/b16_narrow.err
CSGPU_NO_LONGROW=1 timeout 300 $B --batch 16 > $OUT/b16_nolong.json 2> $OUT/b16_nolong.err
timeout 300 $B --batch 8 > $OUT/b8_on.json 2> $OUT/b8_on.err
timeout 300 $B --batch 8 --opt two_product=-1 > $OUT/b8_off.json 2> $OUT/b8_off.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/twoprod/b*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "value %.2f ms/step %.1f" % (d["value"], d["ms_per_step"]), json.dumps(d["config"])[:400])
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
