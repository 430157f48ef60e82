#!/bin/bash
# Round 3, call A: GPU test-suite (incl. the new real-device twins), the driver's bench command (fp64 path = `value`,
# mixed path beside it, each with its roofline), rocprofv3 kernel stats + PMC passes of the ALL-FP64 path at full size.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3a
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 6000 $OUT/bench.json
TAG=r3a_fp64 STEPS=3 BENCH_ARGS="--precond same --host-csr 0" bash tools/gpu_r2_prof.sh > $OUT/prof_fp64.log 2>&1; head -14 $OUT/prof_fp64.log
BENCH_ARGS="--precond same --host-csr 0" bash tools/gpu_pmc.sh > $OUT/pmc_fp64.log 2>&1; tail -16 $OUT/pmc_fp64.log
cp gpurun_out/pmc_bench/pmc_by_kernel.json $OUT/pmc_by_kernel_fp64.json
