#!/bin/bash
# Round 2, final evidence: GPU test-suite, smoke, the driver's bench command, rocprofv3 stats, full-size parity.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2final
rm -rf $OUT; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 7000 $OUT/bench.json
TAG=r2final bash tools/gpu_r2_prof.sh > $OUT/prof.log 2>&1; head -12 $OUT/prof.log
timeout 1500 python tools/full_size_checks.py --out $OUT/parity_10000.json > $OUT/full_size.log 2>&1; tail -c 1500 $OUT/full_size.log
