#!/bin/bash
# Round 4, GPU call K: launch bound of the residual update at K = 32 (build-time knob CSGPU_RUPD_WAVES: default / 3 / 4 waves per
# SIMD), default tile width now 64 at K = 32; per-kernel times from the bench's own events are not enough (only the CG product
# is timed), so rocprofv3 kernel stats of a short run per build.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4k
rm -rf $OUT; mkdir -p $OUT
for V in default rupd3 rupd4; do
  if [ $V = default ]; then unset CSGPU_LIB; else export CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu_$V.so; fi
  timeout 300 python bench.py --steps 6 --warmup 2 --cpu-sample 0 --host-csr 0 --extra-legs 0 2> /dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); m=d.get('mixed_path',{}); print('$V: fp64 ms/16 %.1f roof %.3f | mixed ms/16 %.1f' % (d['ms_per_16_pairs'], d['roofline']['frac'], m.get('ms_per_16_pairs',0)))" | tee -a $OUT/rupd_waves.txt
done
cd /tmp && export TMPDIR=/tmp
for V in default rupd3; do
  if [ $V = default ]; then unset CSGPU_LIB; else export CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu_$V.so; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw_$V -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 --host-csr 0 --extra-legs 0 --compare-steps 0 > /dev/null 2>&1
  find $OUT/raw_$V -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$V.csv \;
  rm -rf $OUT/raw_$V
  echo "$V:"; head -6 $OUT/kernel_stats_$V.csv | cut -c1-150
done
