#!/bin/bash
# Round 6, closing build: the six older fuzzers on the device with new seeds (raster paths now read the column-major copy of the
# raster; lattice_ap_q_kernel with its loads up front; enriched levels on the fused pass)
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6fuzzlast
rm -rf $OUT; mkdir -p $OUT
export CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu.so
timeout 600 python tools/fuzz_polygons.py 120 81 > $OUT/fuzz_polygons.jsonl 2> $OUT/fuzz_polygons.err; tail -1 $OUT/fuzz_polygons.jsonl | cut -c1-300; grep '"ok": false\|error' $OUT/fuzz_polygons.jsonl | head -5 | cut -c1-400
for SEED in 81 82; do
  timeout 500 python tools/fuzz_rasters.py $SEED 120 > $OUT/fuzz_rasters_$SEED.log 2>&1; tail -1 $OUT/fuzz_rasters_$SEED.log | cut -c1-200; grep "EXC\|BAD" $OUT/fuzz_rasters_$SEED.log | head -4 | cut -c1-400
done
FUZZ_MIN=40 FUZZ_MAX=200 timeout 500 python tools/fuzz_rasters.py 83 60 > $OUT/fuzz_rasters_83_large.log 2>&1; tail -1 $OUT/fuzz_rasters_83_large.log | cut -c1-200; grep "EXC\|BAD" $OUT/fuzz_rasters_83_large.log | head -4 | cut -c1-400
timeout 500 python tools/fuzz_networks.py 81 100 > $OUT/fuzz_networks_81.log 2>&1; tail -1 $OUT/fuzz_networks_81.log | cut -c1-200; grep "EXC\|BAD" $OUT/fuzz_networks_81.log | head -4 | cut -c1-400
timeout 500 python tools/fuzz_stream.py 60 81 > $OUT/fuzz_stream.log 2>&1; tail -1 $OUT/fuzz_stream.log | cut -c1-300; grep "EXC\|BAD\|false" $OUT/fuzz_stream.log | head -4 | cut -c1-400
timeout 500 python tools/fuzz_dia25.py 81 40 > $OUT/fuzz_dia25.log 2>&1; tail -1 $OUT/fuzz_dia25.log | cut -c1-300; grep "EXC\|BAD" $OUT/fuzz_dia25.log | head -4 | cut -c1-400
timeout 500 python tools/fuzz_streamed.py 81 60 > $OUT/fuzz_streamed.log 2>&1; tail -1 $OUT/fuzz_streamed.log | cut -c1-300; grep "EXC\|BAD" $OUT/fuzz_streamed.log | head -4 | cut -c1-400
