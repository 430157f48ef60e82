#!/bin/bash
# Round 2, call L: full GPU test-suite, one-to-all hierarchy reuse at 2000^2 (16 focal points), default bench.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2l
rm -rf $OUT; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 600 python tools/onetoall_bench.py 2000 2> $OUT/onetoall.err | tee $OUT/onetoall.jsonl; tail -2 $OUT/onetoall.err
timeout 300 python bench.py --compare-steps 0 --cpu-sample 0 --steps 5 2> $OUT/bench.err | tee $OUT/bench.json | cut -c1-400
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
