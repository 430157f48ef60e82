#!/bin/bash
# Round 5, first GPU call: the WHOLE GPU suite (no -x: every failure is wanted), smoke(), the driver's bench command.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5a
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $OUT/pytest_gpu.log 2>&1; tail -25 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json
