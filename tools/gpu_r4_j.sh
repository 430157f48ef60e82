#!/bin/bash
# Round 4, GPU call J: tile width of the marching kernels at K = 32 (CSGPU_DIA_SEG), geometric network at n = 5e6.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4j
rm -rf $OUT; mkdir -p $OUT
for SEG in 32 48 64 96; do
  CSGPU_DIA_SEG=$SEG timeout 300 python bench.py --steps 6 --warmup 2 --cpu-sample 0 --host-csr 0 --extra-legs 0 2> /dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); m=d.get('mixed_path',{}); print('DIA_SEG $SEG: fp64 ms/16 %.1f roof %.3f (%.2f ms) | mixed ms/16 %.1f roof %.3f' % (d['ms_per_16_pairs'], d['roofline']['frac'], d['roofline']['avg_ms'], m.get('ms_per_16_pairs',0), m.get('roofline',{}).get('frac',0)))" | tee -a $OUT/dia_seg_k32.txt
done
timeout 900 python tools/network_locality_bench.py 5000000 > $OUT/network_geometric_5e6.jsonl 2> $OUT/network.err; cut -c1-600 $OUT/network_geometric_5e6.jsonl; tail -2 $OUT/network.err
