#!/bin/bash
# Round 2, call Q: waves-per-SIMD launch bounds of the marching kernels (A/B builds).
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2q
rm -rf $OUT; mkdir -p $OUT
B="python bench.py --compare-steps 0 --cpu-sample 0 --steps 5"
timeout 200 $B > $OUT/cg4_rupd3.json 2> $OUT/cg4_rupd3.err
CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu_nobound.so timeout 200 $B > $OUT/nobound.json 2> $OUT/nobound.err
CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu_rupd3.so timeout 200 $B > $OUT/cg_free_rupd3.json 2> $OUT/cg_free_rupd3.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2q/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "value %.2f ms/step %.1f pcg_ms %.1f cg_prod_ms %.3f frac %.3f iters %.2f" % (d["value"], d["ms_per_step"], d["pcg_device_ms_per_step"], d["roofline"]["avg_ms"], d["roofline"]["frac"], d["iters_mean"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e, open(f[:-5]+".err").read()[-500:])
PY
TAG=r2q bash tools/gpu_r2_prof.sh 2>&1 | head -8
