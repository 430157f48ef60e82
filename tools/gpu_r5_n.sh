#!/bin/bash
# Round 5: the raster and stream fuzzers on the DEVICE build with the coarse-space enrichment on (aggressive threshold 0.15
# and the default), rasters 60-220 cells a side.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5n
rm -rf $OUT; mkdir -p $OUT
export CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu.so
for SEED in 61 62; do
  CSGPU_ENRICH_TAU=0.15 FUZZ_MIN=60 FUZZ_MAX=220 timeout 500 python tools/fuzz_rasters.py $SEED 150 > $OUT/fuzz_rasters_tau015_$SEED.log 2>&1; tail -1 $OUT/fuzz_rasters_tau015_$SEED.log | cut -c1-200; grep "EXC\|BAD" $OUT/fuzz_rasters_tau015_$SEED.log | head -4 | cut -c1-400
done
FUZZ_MIN=60 FUZZ_MAX=220 timeout 500 python tools/fuzz_rasters.py 63 150 > $OUT/fuzz_rasters_default_63.log 2>&1; tail -1 $OUT/fuzz_rasters_default_63.log | cut -c1-200; grep "EXC\|BAD" $OUT/fuzz_rasters_default_63.log | head -4 | cut -c1-400
FUZZ_MIN=60 FUZZ_MAX=200 timeout 500 python tools/fuzz_stream.py 60 71 > $OUT/fuzz_stream_71.log 2>&1; tail -2 $OUT/fuzz_stream_71.log | cut -c1-300
