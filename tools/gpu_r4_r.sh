#!/bin/bash
# Round 4, last GPU call: the whole GPU suite on the closing build (with the two regression tests of the device fuzz findings),
# then the raster / network fuzzers at larger sizes again, time-boxed.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4r
rm -rf $OUT; mkdir -p $OUT
timeout 420 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
export CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu.so
FUZZ_MIN=60 FUZZ_MAX=200 timeout 150 python tools/fuzz_rasters.py 51 30 > $OUT/fuzz_rasters_big_51.log 2>&1; tail -1 $OUT/fuzz_rasters_big_51.log | cut -c1-200; grep "EXC\|BAD" $OUT/fuzz_rasters_big_51.log | head -6 | cut -c1-400
FUZZ_MIN=60 FUZZ_MAX=200 timeout 100 python tools/fuzz_rasters.py 52 30 > $OUT/fuzz_rasters_big_52.log 2>&1; tail -1 $OUT/fuzz_rasters_big_52.log | cut -c1-200; grep "EXC\|BAD" $OUT/fuzz_rasters_big_52.log | head -6 | cut -c1-400
FUZZ_MIN=2000 FUZZ_MAX=20000 timeout 100 python tools/fuzz_networks.py 52 25 > $OUT/fuzz_networks_big_52.log 2>&1; tail -1 $OUT/fuzz_networks_big_52.log | cut -c1-200; grep "EXC\|BAD" $OUT/fuzz_networks_big_52.log | head -6 | cut -c1-400
