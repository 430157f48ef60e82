#!/bin/bash
# Round 6, fifth GPU call: the whole device suite on the build with the knobs refactor, the node-space residuals and the new
# tests; smoke; the six fuzzers on the device build (new seeds).
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6e
rm -rf $OUT; mkdir -p $OUT
export CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu.so
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest_gpu.log 2>&1; tail -14 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 600 python tools/fuzz_polygons.py 120 61 > $OUT/fuzz_polygons.jsonl 2> $OUT/fuzz_polygons.err; tail -1 $OUT/fuzz_polygons.jsonl | cut -c1-300; grep '"ok": false\|error' $OUT/fuzz_polygons.jsonl | head -5 | cut -c1-400
for SEED in 61 62; do
  timeout 500 python tools/fuzz_rasters.py $SEED 120 > $OUT/fuzz_rasters_$SEED.log 2>&1; tail -1 $OUT/fuzz_rasters_$SEED.log | cut -c1-200; grep "EXC\|BAD" $OUT/fuzz_rasters_$SEED.log | head -4 | cut -c1-400
  timeout 500 python tools/fuzz_networks.py $SEED 120 > $OUT/fuzz_networks_$SEED.log 2>&1; tail -1 $OUT/fuzz_networks_$SEED.log | cut -c1-200; grep "EXC\|BAD" $OUT/fuzz_networks_$SEED.log | head -4 | cut -c1-400
done
timeout 500 python tools/fuzz_stream.py 60 61 > $OUT/fuzz_stream.log 2>&1; tail -1 $OUT/fuzz_stream.log | cut -c1-300; grep "EXC\|BAD\|false" $OUT/fuzz_stream.log | head -4 | cut -c1-400
timeout 500 python tools/fuzz_dia25.py 61 40 > $OUT/fuzz_dia25.log 2>&1; tail -1 $OUT/fuzz_dia25.log | cut -c1-300; grep "EXC\|BAD" $OUT/fuzz_dia25.log | head -4 | cut -c1-400
timeout 500 python tools/fuzz_streamed.py 61 60 > $OUT/fuzz_streamed.log 2>&1; tail -1 $OUT/fuzz_streamed.log | cut -c1-300; grep "EXC\|BAD" $OUT/fuzz_streamed.log | head -4 | cut -c1-400
