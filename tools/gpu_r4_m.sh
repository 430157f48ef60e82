#!/bin/bash
# Round 4, GPU call M: closing numbers on the final build -- NODATA / blobs / all-valid at 10000^2 (K = 32, batch mode), the
# fixed 100-pair job of configs[2] and the 1000-pair fp32 job of configs[3] on ONE GPU (T_1 of the strong-scaling arithmetic),
# bench.py --precision single.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4m
rm -rf $OUT; mkdir -p $OUT
MODES=batch PBS=0,4 BATCHES=32 PAIRS=96 timeout 900 python tools/stream_bench.py 10000 valid,holes15,blobs15 > $OUT/rasters_10000_k32.jsonl 2> $OUT/rasters.err
python - $OUT/rasters_10000_k32.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print("  %-8s pb%d K%-2d ms/16 %.1f iters %.2f/%d" % (d["case"], d["precond_bytes"], d["batch"], d["ms_per_16_pairs"], d["iters_mean"], d["iters_max"]))
PY
timeout 300 python bench.py --scaling strong --pairs 100 > $OUT/strong_100_fp64_1gpu.json 2> $OUT/strong.err; python -c "
import json; d=json.loads(open('$OUT/strong_100_fp64_1gpu.json').read().strip().splitlines()[-1]); print('strong 100 pairs fp64, 1 GPU: job %.2fs setup %.2fs per batch %.3fs value %.1f' % (d['job_s'], d['rank_setup_s'][0], d['per_batch_s_rank0'], d['value']))"
timeout 300 python bench.py --scaling strong --pairs 1000 --precision single > $OUT/strong_1000_fp32_1gpu.json 2>> $OUT/strong.err; python -c "
import json; d=json.loads(open('$OUT/strong_1000_fp32_1gpu.json').read().strip().splitlines()[-1]); print('strong 1000 pairs fp32, 1 GPU: job %.2fs setup %.2fs per batch %.3fs value %.1f' % (d['job_s'], d['rank_setup_s'][0], d['per_batch_s_rank0'], d['value']))"
timeout 400 python bench.py --precision single --steps 10 --warmup 2 --host-csr 0 > $OUT/bench_fp32_k32.json 2> $OUT/bench_fp32.err; python -c "
import json; d=json.loads(open('$OUT/bench_fp32_k32.json').read().strip().splitlines()[-1]); print('bench fp32: value %.1f ms/16 %.1f iters %.2f roof %.3f parity %s' % (d['value'], d['ms_per_16_pairs'], d['iters_mean'], d['roofline']['frac'], d.get('parity',{}).get('max_rel_err_vs_oracle')))"
