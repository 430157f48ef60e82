#!/bin/bash
# Round 6, sixth GPU call: re-validation after the fuzz findings (components ignore zero-weight entries; stream identity at
# one width) and the expander probe: the device suite, the polygon / stream fuzzers on the seeds that reported + new ones, the
# network workload and the driver's bench command.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6f
rm -rf $OUT; mkdir -p $OUT
export CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu.so
timeout 1800 python -m pytest tests -m gpu -q --durations=5 > $OUT/pytest_gpu.log 2>&1; tail -10 $OUT/pytest_gpu.log
timeout 600 python tools/fuzz_polygons.py 120 61 > $OUT/fuzz_polygons_61.jsonl 2> $OUT/fp.err; tail -1 $OUT/fuzz_polygons_61.jsonl | cut -c1-300
timeout 600 python tools/fuzz_polygons.py 150 63 > $OUT/fuzz_polygons_63.jsonl 2>> $OUT/fp.err; tail -1 $OUT/fuzz_polygons_63.jsonl | cut -c1-300; grep '"ok": false\|error' $OUT/fuzz_polygons_63.jsonl | head -4 | cut -c1-400
timeout 500 python tools/fuzz_stream.py 60 61 > $OUT/fuzz_stream_61.log 2>&1; tail -1 $OUT/fuzz_stream_61.log | cut -c1-300
timeout 500 python tools/fuzz_stream.py 60 64 > $OUT/fuzz_stream_64.log 2>&1; tail -1 $OUT/fuzz_stream_64.log | cut -c1-300; grep 'false\|error' $OUT/fuzz_stream_64.log | head -3 | cut -c1-400
timeout 500 python tools/fuzz_networks.py 65 150 > $OUT/fuzz_networks_65.log 2>&1; tail -1 $OUT/fuzz_networks_65.log | cut -c1-200
unset CSGPU_LIB
timeout 600 python bench.py --workload network --gpus 1 --steps 3 --warmup 1 > $OUT/bench_network.json 2> $OUT/bench_network.err; python -c "
import json;d=json.load(open('$OUT/bench_network.json'))
print('network workload', d['value'], d['solve_only_sources_per_s'], d['value_device_rank0'], d['setup_s'], d['setup_device_s'], d['iters_mean'], d['parity']['ok'])"
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; tail -3 $OUT/bench.time
python -c "
import json;d=json.load(open('$OUT/bench.json'))
print('value', d['value'], 'job', d.get('value_job'), 'mixed', d.get('value_mixed'), 'roofline', d['roofline']['frac'], d['roofline'].get('traffic_over_algorithmic'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline'].get('measured'))
print('net', {k:d['config4_network'].get(k) for k in ('value','value_device','setup_s','setup_device_s','solve_s_all_sources','iters_mean')}, d['config4_network']['parity']['ok'])
print('geo', {k:d['network_geometric'].get(k) for k in ('value','value_device','iters_mean','levels')}, d['network_geometric']['parity']['ok'])
print('nodata', d['nodata15']['ms_per_16_pairs'], d['nodata15']['iters_mean'], 'fp32', d['config3_fp32']['value'])
print('parity', d['parity']['max_rel_err_vs_oracle'])
"
