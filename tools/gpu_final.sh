#!/bin/bash
# Round-end evidence run: GPU tests, smoke, default bench, rocprofv3 kernel stats, PMC passes. Outputs -> gpurun_out/final/
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/final
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cat $OUT/bench_default.json | cut -c1-300
BENCH_ARGS="--compare-steps 0" STEPS=3 bash tools/gpu_prof.sh > $OUT/prof_stdout.txt 2>&1; head -12 $OUT/prof_stdout.txt
cp gpurun_out/prof/kernel_stats.csv $OUT/kernel_stats.csv
bash tools/gpu_pmc.sh > $OUT/pmc_stdout.txt 2>&1; grep -E "spmv_kernel<double, 16, 0, true|spmv_kernel<float, 16, 0, true|spmm_longrow_kernel<float, 16, 32|cg_update" $OUT/pmc_stdout.txt
cp gpurun_out/pmc_bench/pmc_by_kernel.json $OUT/pmc_by_kernel.json
