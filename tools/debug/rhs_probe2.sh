#!/bin/bash
OUT=gpurun_out/r3d; mkdir -p $OUT
run() { env "$@" timeout 120 python tools/debug/rhs_probe2.py >> $OUT/rhs_probe2.jsonl 2>> $OUT/rhs_probe2.err; }
rm -f $OUT/rhs_probe2.jsonl
run KNOB=base FINGERPRINT=1
run KNOB=pb0 PB=0 FINGERPRINT=1
run KNOB=nodefl CSGPU_NO_DEFLATION=1
run KNOB=notailproj CSGPU_NO_TAIL_PROJECTION=1
run KNOB=notail CSGPU_TAIL_ROWS=0
run KNOB=nopieces CSGPU_NO_TILE_PIECES=1
run KNOB=nodirectlattice CSGPU_NO_DIRECT_LATTICE=1
run KNOB=nocheb CSGPU_COARSE_CHEBYSHEV=0
run KNOB=kernelref CSGPU_KERNEL_GAIN_REF=1
cut -c1-300 $OUT/rhs_probe2.jsonl
