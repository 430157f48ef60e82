#!/bin/bash
# Standard GPU session: microbench, parity tests, headline bench. Outputs under gpurun_out/.
mkdir -p gpurun_out
python tools/spmm_bench.py 10000 double 1,8 2>&1 | tail -1
python tools/spmm_bench.py 10000 single 1,8 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
timeout 900 python bench.py ${BENCH_ARGS} > gpurun_out/bench_10000.json 2> gpurun_out/bench_10000.err; echo "rc=$?"
cat gpurun_out/bench_10000.json; tail -3 gpurun_out/bench_10000.err
