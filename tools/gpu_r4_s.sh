#!/bin/bash
# Round 4, GPU call S: the streaming fuzzer on the device build (batch / stream / adaptive on one handle, K = 8 / 16 / 32), small
# and larger rasters, time-boxed.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4s
rm -rf $OUT; mkdir -p $OUT
export CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu.so
timeout 120 python tools/fuzz_stream.py 150 3 > $OUT/fuzz_stream_small.jsonl 2> $OUT/fuzz_stream_small.err; tail -1 $OUT/fuzz_stream_small.jsonl; grep '"ok": false\|error' $OUT/fuzz_stream_small.jsonl | head -5 | cut -c1-400
FUZZ_MIN=150 FUZZ_MAX=700 timeout 150 python tools/fuzz_stream.py 60 4 > $OUT/fuzz_stream_big.jsonl 2> $OUT/fuzz_stream_big.err; tail -1 $OUT/fuzz_stream_big.jsonl; grep '"ok": false\|error' $OUT/fuzz_stream_big.jsonl | head -5 | cut -c1-400
