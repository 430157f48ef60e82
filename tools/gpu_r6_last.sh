#!/bin/bash
# Round 6, last device call: the library as __graft_entry__.build() leaves it (HEAD) -- device suite + smoke
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6last
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
