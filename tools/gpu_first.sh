#!/bin/bash
# First GPU session: parity tests, small + headline bench, rocprofv3 kernel stats. Outputs under gpurun_out/.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/device.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --size 1000 --steps 4 --warmup 1 --cpu-sample 0 > gpurun_out/bench_1000.json 2> gpurun_out/bench_1000.err; echo "rc=$?"
cat gpurun_out/bench_1000.json; tail -5 gpurun_out/bench_1000.err
timeout 900 python bench.py > gpurun_out/bench_10000.json 2> gpurun_out/bench_10000.err; echo "rc=$?"
cat gpurun_out/bench_10000.json; tail -5 gpurun_out/bench_10000.err
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_bench.err; echo "rc=$?"
cd $GRAFT_REPO_ROOT
ls -la gpurun_out/prof_r1 | head; find gpurun_out/prof_r1 -name "*stats*" | head
# keep the merge-back small: drop the raw kernel trace, keep stats
find gpurun_out/prof_r1 -name "*kernel_trace*" -size +20M -delete
for f in $(find gpurun_out/prof_r1 -name "*kernel_stats*"); do head -40 $f; done
