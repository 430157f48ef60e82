#!/bin/bash
# Round 4, GPU call F: how wide must a long polygon be for the lattice path? 200 strips of up to 80 cells, widths 2..8, lattice
# path against the merged CSR path
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4f
mkdir -p $OUT
for W in 2 3 4 5 6 8; do
  for MODE in lattice csr; do
    if [ $MODE = csr ]; then export CSGPU_NO_POLY_LATTICE=1; else unset CSGPU_NO_POLY_LATTICE; fi
    POLY_WIDTH=$W POLY_MAX=80 POLY_SHAPE=lines PBS=0 timeout 200 python tools/polygon_bench.py 5000 200 2> /dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('200 strips of width $W, $MODE: ms/batch %.1f iters %.2f/%d setup %.2f (%s)' % (d['ms_per_batch'], d['iters_mean'], d['iters_max'], d['setup_wall_s'], d['case']))" | tee -a $OUT/strip_width_sweep.txt
  done
done
unset CSGPU_NO_POLY_LATTICE
for MODE in lattice csr; do
  if [ $MODE = csr ]; then export CSGPU_NO_POLY_LATTICE=1; else unset CSGPU_NO_POLY_LATTICE; fi
  POLY_MAX=60 PBS=0 timeout 200 python tools/polygon_bench.py 5000 300 2> /dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('300 overlapping rectangles <= 60, $MODE: ms/batch %.1f iters %.2f/%d setup %.2f' % (d['ms_per_batch'], d['iters_mean'], d['iters_max'], d['setup_wall_s']))" | tee -a $OUT/strip_width_sweep.txt
done
