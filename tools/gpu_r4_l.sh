#!/bin/bash
# Round 4, GPU call L: residual update with its r loads in flight one column ahead (libcsgpu_rpre.so) against the build before
# it (libcsgpu.so): bench A/B twice each + rocprofv3 kernel stats.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4l
rm -rf $OUT; mkdir -p $OUT
for V in before rpre before rpre; do
  if [ $V = before ]; then unset CSGPU_LIB; else export CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu_$V.so; fi
  timeout 300 python bench.py --steps 6 --warmup 2 --cpu-sample 0 --host-csr 0 --extra-legs 0 2> /dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); m=d.get('mixed_path',{}); print('$V: fp64 ms/16 %.1f roof %.3f parity-iters %.2f | mixed ms/16 %.1f' % (d['ms_per_16_pairs'], d['roofline']['frac'], d['iters_mean'], m.get('ms_per_16_pairs',0)))" | tee -a $OUT/rpre_ab.txt
done
cd /tmp && export TMPDIR=/tmp
for V in before rpre; do
  if [ $V = before ]; then unset CSGPU_LIB; else export CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu_$V.so; fi
  for P in same fp32; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw_$V$P -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 --host-csr 0 --extra-legs 0 --compare-steps 0 --precond $P > /dev/null 2>&1
  find $OUT/raw_$V$P -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_${V}_$P.csv \;
  rm -rf $OUT/raw_$V$P
  echo "$V $P:"; grep "32, 3>" $OUT/kernel_stats_${V}_$P.csv | cut -c1-170
  done
done
