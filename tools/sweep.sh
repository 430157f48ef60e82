#!/bin/bash
# tuning sweep on the headline workload (short, individually time-limited runs)
mkdir -p gpurun_out
: > gpurun_out/sweep.txt
IFS=";" read -ra LIST <<< "$SWEEP"
for o in "${LIST[@]}"; do
  timeout 90 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --compare-steps 0 --opt itmax=120 $o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$o', '| levels', d['config']['levels'], 'opcx', round(d['config']['operator_complexity'],3), 'iters', round(d['iters_mean'],2), d['iters_max'], 'ms/step', round(d['ms_per_step'],1), 'pairs/s', round(d['solve_only_pairs_per_s'],2))" >> gpurun_out/sweep.txt 2>&1 || echo "$o FAILED/TIMEOUT" >> gpurun_out/sweep.txt
done
cat gpurun_out/sweep.txt
