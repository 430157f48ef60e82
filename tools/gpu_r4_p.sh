#!/bin/bash
# Round 4, GPU call P: the three fuzzers (polygons; rasters and networks of round 3) on the REAL device build -- round 3 ran
# them on the emulator build only, and the device found what the emulator could not (a 696-node star over six decades on
# an fp32 hierarchy: csgpu.hip, setup_from_host).
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4p
rm -rf $OUT; mkdir -p $OUT
export CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu.so
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "contrast_triggered" 2>&1 | tail -2
timeout 900 python tools/fuzz_polygons.py 200 ${SEEDP:-12} > $OUT/fuzz_polygons_gpu.jsonl 2> $OUT/fuzz_polygons.err; tail -1 $OUT/fuzz_polygons_gpu.jsonl; grep '"ok": false\|error' $OUT/fuzz_polygons_gpu.jsonl | head -5 | cut -c1-400
for SEED in ${SEEDS:-41 42 43}; do
  timeout 600 python tools/fuzz_rasters.py $SEED 150 > $OUT/fuzz_rasters_gpu_$SEED.log 2>&1; tail -1 $OUT/fuzz_rasters_gpu_$SEED.log | cut -c1-200; grep "EXC\|BAD" $OUT/fuzz_rasters_gpu_$SEED.log | head -4 | cut -c1-400
  timeout 600 python tools/fuzz_networks.py $SEED 150 > $OUT/fuzz_networks_gpu_$SEED.log 2>&1; tail -1 $OUT/fuzz_networks_gpu_$SEED.log | cut -c1-200; grep "EXC\|BAD" $OUT/fuzz_networks_gpu_$SEED.log | head -4 | cut -c1-400
done
