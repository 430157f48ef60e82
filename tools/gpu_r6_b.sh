#!/bin/bash
# Round 6, second GPU call: (1) degree of the single-level polynomial preconditioner x batch width on the configs[4] network;
# (2) the 100-pair job with the batch width picked per batch against every batch at the call's width (--opt fixed_k=1);
# (3) residual update at 2 waves per SIMD (build -DCSGPU_RUPD_WAVES=2) against 1.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6b
rm -rf $OUT; mkdir -p $OUT
timeout 900 python tools/network_sweeps.py > $OUT/network_sweeps.jsonl 2> $OUT/network_sweeps.err; cat $OUT/network_sweeps.jsonl | cut -c1-420
SHORT="--steps 4 --warmup 2 --extra-legs 0 --cpu-sample 0 --compare-steps 0 --host-csr 0 --pmc-live 0"
timeout 600 python bench.py $SHORT > $OUT/bench_perbatch_k.json 2> $OUT/e1; python -c "import json;d=json.load(open('$OUT/bench_perbatch_k.json'));print('per-batch K', d['ms_per_16_pairs'], d['value'], d['job_100_pairs'])"
timeout 600 python bench.py $SHORT --opt fixed_k=1 > $OUT/bench_fixed_k.json 2> $OUT/e2; python -c "import json;d=json.load(open('$OUT/bench_fixed_k.json'));print('fixed K', d['ms_per_16_pairs'], d['value'], d['job_100_pairs'])"
CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu_rupd2.so timeout 600 python bench.py $SHORT > $OUT/bench_rupd_waves2.json 2> $OUT/e3; python -c "import json;d=json.load(open('$OUT/bench_rupd_waves2.json'));print('rupd waves 2', d['ms_per_16_pairs'], d['value'], d['roofline']['avg_ms'])"
timeout 600 python bench.py $SHORT > $OUT/bench_rupd_waves1.json 2> $OUT/e4; python -c "import json;d=json.load(open('$OUT/bench_rupd_waves1.json'));print('rupd waves 1', d['ms_per_16_pairs'], d['value'], d['roofline']['avg_ms'])"
