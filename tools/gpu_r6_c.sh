#!/bin/bash
# Round 6, third GPU call: the pins of VERDICT r5 item 8 -- (a) the fp32 fixture at 10000^2, (b) a real host matrix with
# >= 2^31 stored entries through csgpu_setup, (c) the driver's bench command with the CPU baseline measured at the full size
# in the background -- and BASELINE configs[4] as its own workload with the new defaults (batch 32, one Jacobi sweep).
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6c
rm -rf $OUT; mkdir -p $OUT
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; tail -3 $OUT/bench.time
python -c "
import json;d=json.load(open('$OUT/bench.json'))
print('value', d['value'], 'job', d.get('value_job'), 'mixed', d.get('value_mixed'))
print('cpu', {k:v for k,v in d['cpu_baseline'].items() if k not in ('full_size','sample','bounded_sample','measured_full_size')})
print('legs', d['leg_seconds'])
print('net', {k:d['config4_network'].get(k) for k in ('value','value_device','batch','iters_mean','solve_s_all_sources','solve_device_s_all_sources','preconditioner')}, d['config4_network']['parity']['ok'], d['config4_network']['roofline']['frac'])
print('geo', {k:d['network_geometric'].get(k) for k in ('value','value_device','iters_mean')}, d['network_geometric']['roofline']['frac'])
print('nodata', d['nodata15']['ms_per_16_pairs'], d['nodata15']['iters_mean'])
"
timeout 600 python bench.py --workload network --gpus 1 --steps 3 --warmup 1 > $OUT/bench_network.json 2> $OUT/bench_network.err; python -c "
import json;d=json.load(open('$OUT/bench_network.json'))
print('network workload', d['value'], d['solve_only_sources_per_s'], d['value_device_rank0'], d['iters_mean'], d['parity'], d['roofline']['frac'])"
timeout 1500 python tools/full_size_fp32.py > $OUT/fp32.log 2>&1; tail -2 $OUT/fp32.log | cut -c1-600
timeout 1500 python tools/host_csr_2e31.py > $OUT/host2e31.log 2>&1; tail -2 $OUT/host2e31.log | cut -c1-1200
cp profiles/r6_parity16_10000_fp32.json profiles/r6_host_csr_2e31.json tests/golden/full_size_10000_fp32.json $OUT/ 2>/dev/null
