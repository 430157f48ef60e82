#!/bin/bash
# Round 2, call J: PMC traffic of the current kernels (separate FETCH_SIZE / WRITE_SIZE passes), all-fp32 handle (BASELINE config 4 precision).
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2j
rm -rf $OUT $GRAFT_REPO_ROOT/gpurun_out/pmc_bench; mkdir -p $OUT
bash tools/gpu_pmc.sh > $OUT/pmc.log 2>&1; tail -16 $OUT/pmc.log
B="python bench.py --compare-steps 0 --cpu-sample 0 --steps 5"
timeout 200 $B --precision single > $OUT/fp32_default_tol.json 2> $OUT/fp32_default_tol.err
timeout 200 $B --precision single --opt rtol=1e-5 --opt atol=0 > $OUT/fp32_rtol1e-5.json 2> $OUT/fp32_rtol1e-5.err
for f in fp32_default_tol fp32_rtol1e-5; do tail -c 1800 $OUT/$f.json; echo; tail -2 $OUT/$f.err; done
