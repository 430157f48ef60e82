#!/bin/bash
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/nt
rm -rf $OUT; mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
if ! grep -q " passed" $OUT/pytest_gpu.log || grep -q "failed" $OUT/pytest_gpu.log; then echo "GPU TESTS FAILED"; exit 1; fi
B="python bench.py --compare-steps 0 --cpu-sample 0 --steps 4 --opt itmax=200"
CSGPU_VERBOSE=1 timeout 300 $B --batch 16 > $OUT/b16_nt.json 2> $OUT/b16_nt.err; grep csgpu: $OUT/b16_nt.err
CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu_nont.so timeout 300 $B --batch 16 > $OUT/b16_plain.json 2> $OUT/b16_plain.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/nt/b*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "value %.2f ms/step %.1f spmm_ms %s" % (d["value"], d["ms_per_step"], d["roofline"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
BENCH_ARGS="--opt itmax=200" bash tools/gpu_pmc.sh > $OUT/pmc_stdout.txt 2>&1; grep -E "spmv_kernel<double, 16, 0, true|spmv_kernel<float, 16, 0, true|spmm_longrow_kernel<float, 16, 32|cg_update_r" $OUT/pmc_stdout.txt
