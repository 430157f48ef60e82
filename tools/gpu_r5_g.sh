#!/bin/bash
# Round 5: the CPU baseline measured at the full size (host cores only) and the driver's cost of fresh device memory.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5g
rm -rf $OUT; mkdir -p $OUT
nproc; free -g | head -2
for m in one many; do timeout 120 python tools/alloc_probe2.py 40 $m >> $OUT/alloc_probe.jsonl 2>> $OUT/alloc.err; done
timeout 120 python tools/alloc_probe2.py 100 one >> $OUT/alloc_probe.jsonl 2>> $OUT/alloc.err
cat $OUT/alloc_probe.jsonl
THREADS=16 timeout 900 python tools/cpu_full_size.py 10000 $OUT/cpu_baseline_full_size.json 2> $OUT/cpu.err | cut -c1-600
tail -n 3 $OUT/cpu.err $OUT/alloc.err
