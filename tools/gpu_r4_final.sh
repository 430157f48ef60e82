#!/bin/bash
# Round 4, evidence run of the final build: GPU suite, smoke, the driver's bench command, rocprofv3 kernel stats + PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate passes) of the all-fp64 path and of the mixed path at full size, batches of 32.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4final
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > $OUT/pytest_gpu.log 2>&1; tail -12 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json
TAG=r4final_fp64 STEPS=3 BENCH_ARGS="--precond same --host-csr 0 --extra-legs 0" bash tools/gpu_r2_prof.sh > $OUT/prof_fp64.log 2>&1; head -10 $OUT/prof_fp64.log
cp gpurun_out/prof_r4final_fp64/kernel_stats.csv $OUT/kernel_stats_fp64.csv
cp gpurun_out/prof_r4final_fp64/kernel_stats_fullsize.json $OUT/kernel_stats_fp64_fullsize.json
BENCH_ARGS="--precond same --host-csr 0 --extra-legs 0" bash tools/gpu_pmc.sh > $OUT/pmc_fp64.log 2>&1; tail -8 $OUT/pmc_fp64.log
cp gpurun_out/pmc_bench/pmc_by_kernel.json $OUT/pmc_by_kernel_fp64.json
BENCH_ARGS="--precond fp32 --host-csr 0 --extra-legs 0" bash tools/gpu_pmc.sh > $OUT/pmc_mixed.log 2>&1; tail -8 $OUT/pmc_mixed.log
cp gpurun_out/pmc_bench/pmc_by_kernel.json $OUT/pmc_by_kernel_mixed.json
