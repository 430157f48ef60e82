#!/bin/bash
# Round 4, GPU call N: the fixed jobs of configs[2] (100 pairs, fp64) and configs[3] (1000 pairs, fp32) on ONE GPU, cold and
# warm (T_1 of the strong-scaling arithmetic in DESIGN.md section 6).
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4n
rm -rf $OUT; mkdir -p $OUT
timeout 300 python bench.py --scaling strong --pairs 100 > $OUT/strong_100_fp64_1gpu.json 2> $OUT/strong.err; python -c "
import json; d=json.loads(open('$OUT/strong_100_fp64_1gpu.json').read().strip().splitlines()[-1]); print('strong 100 pairs fp64, 1 GPU: job %.2fs (cold %.2fs) setup %.2fs per batch %.3fs value %.1f' % (d['job_s'], d['job_cold_s'], d['rank_setup_s'][0], d['per_batch_s_rank0'], d['value']))"
timeout 300 python bench.py --scaling strong --pairs 13 --batch 16 > $OUT/strong_13_fp64_1gpu.json 2>> $OUT/strong.err; python -c "
import json; d=json.loads(open('$OUT/strong_13_fp64_1gpu.json').read().strip().splitlines()[-1]); print('a rank share of 13 pairs fp64: job %.2fs (cold %.2fs) setup %.2fs' % (d['job_s'], d['job_cold_s'], d['rank_setup_s'][0]))"
timeout 300 python bench.py --scaling strong --pairs 1000 --precision single > $OUT/strong_1000_fp32_1gpu.json 2>> $OUT/strong.err; python -c "
import json; d=json.loads(open('$OUT/strong_1000_fp32_1gpu.json').read().strip().splitlines()[-1]); print('strong 1000 pairs fp32, 1 GPU: job %.2fs (cold %.2fs) setup %.2fs per batch %.3fs value %.1f' % (d['job_s'], d['job_cold_s'], d['rank_setup_s'][0], d['per_batch_s_rank0'], d['value']))"
timeout 300 python bench.py --scaling strong --pairs 125 --precision single > $OUT/strong_125_fp32_1gpu.json 2>> $OUT/strong.err; python -c "
import json; d=json.loads(open('$OUT/strong_125_fp32_1gpu.json').read().strip().splitlines()[-1]); print('a rank share of 125 pairs fp32: job %.2fs (cold %.2fs) setup %.2fs' % (d['job_s'], d['job_cold_s'], d['rank_setup_s'][0]))"
