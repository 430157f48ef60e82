#!/bin/bash
# Round 5, last seconds of the GPU budget: three fuzz cases that reported "did not converge" with a tiny residual, with the
# enrichment off and on.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5o
mkdir -p $OUT
export CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu.so
for c in "63 27" "63 101" "62 46"; do timeout 40 python tools/debug/fuzz_case_repro.py $c >> $OUT/repro.log 2>&1; done
cat $OUT/repro.log
