#!/bin/bash
# Round 4, GPU call A: the new at-scale tests (configs[3] fp32 at 5000^2 with the reference's defaults, configs[4] networks
# at n = 1e6), the 16-pair full-size fixture from the tight CPU oracle, the fp32 bench line.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4a
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -s > $OUT/pytest_scale.log 2>&1; tail -15 $OUT/pytest_scale.log
timeout 200 python bench.py --precision single --steps 5 --warmup 2 --host-csr 0 > $OUT/bench_fp32.json 2> $OUT/bench_fp32.err; tail -c 1500 $OUT/bench_fp32.json
timeout 900 python tools/full_size_checks.py --pairs 16 --skip-host-csr 1 --out $OUT/parity16_10000.json --fixture $OUT/full_size_10000.json > $OUT/full_size.log 2> $OUT/full_size.err; tail -c 600 $OUT/full_size.log; tail -5 $OUT/full_size.err
