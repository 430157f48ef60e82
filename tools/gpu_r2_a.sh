#!/bin/bash
# Round 2, call A: single-wave SpMM variant vs the default kernels (bench only, no pytest). Outputs -> gpurun_out/r2a/
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2a
rm -rf $OUT; mkdir -p $OUT
B="python bench.py --compare-steps 0 --cpu-sample 0 --steps 5"
for k in 16 8; do
  timeout 200 $B --batch $k > $OUT/default_k$k.json 2> $OUT/default_k$k.err
  CSGPU_WAVE_SPMM=1 timeout 200 $B --batch $k > $OUT/wave_k$k.json 2> $OUT/wave_k$k.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2a/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "value %.2f ms/step %.1f cg_spmm_ms %.3f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_ms"], d["roofline"]["frac"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
