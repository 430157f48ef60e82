#!/bin/bash
# Round 4, GPU call Q: the fuzzers on the device build at LARGER sizes (rasters 60..220 cells a side, networks 2000..30000
# nodes, polygon rasters 60..200 a side): several levels, lattice level 1, K-wide batches -- what the small cases cannot reach.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4q
rm -rf $OUT; mkdir -p $OUT
export CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu.so
FUZZ_MIN=60 FUZZ_MAX=200 timeout 900 python tools/fuzz_polygons.py 80 21 > $OUT/fuzz_polygons_big.jsonl 2> $OUT/fuzz_polygons_big.err; tail -1 $OUT/fuzz_polygons_big.jsonl; grep '"ok": false\|error' $OUT/fuzz_polygons_big.jsonl | head -5 | cut -c1-400
FUZZ_MIN=60 FUZZ_MAX=220 timeout 900 python tools/fuzz_rasters.py 51 80 > $OUT/fuzz_rasters_big.log 2>&1; tail -1 $OUT/fuzz_rasters_big.log | cut -c1-200; grep "EXC\|BAD" $OUT/fuzz_rasters_big.log | head -5 | cut -c1-500
FUZZ_MIN=2000 FUZZ_MAX=30000 timeout 900 python tools/fuzz_networks.py 51 60 > $OUT/fuzz_networks_big.log 2>&1; tail -1 $OUT/fuzz_networks_big.log | cut -c1-200; grep "EXC\|BAD" $OUT/fuzz_networks_big.log | head -5 | cut -c1-500
