#!/bin/bash
# After the near-kernel fix: one-to-all on the shared hierarchy (2000^2, 16 points), network config 5, all-fp32 handle.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2recheck
rm -rf $OUT; mkdir -p $OUT
timeout 600 python tools/onetoall_bench.py 2000 > $OUT/onetoall.jsonl 2> $OUT/onetoall.err; tail -1 $OUT/onetoall.jsonl | cut -c1-600
CSGPU_NO_DEFLATION=1 timeout 600 python tools/onetoall_bench.py 2000 > $OUT/onetoall_nodefl.jsonl 2> $OUT/onetoall_nodefl.err; tail -1 $OUT/onetoall_nodefl.jsonl | cut -c1-600
NFOCAL=16 timeout 900 python tools/network_bench.py 5000000 16 --shared > $OUT/net5.jsonl 2> $OUT/net5.err; tail -1 $OUT/net5.jsonl | cut -c1-600
timeout 600 python bench.py --precision single --compare-steps 0 --cpu-sample 0 --steps 5 --host-csr 0 > $OUT/fp32.json 2> $OUT/fp32.err
python - $OUT/fp32.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("fp32 handle", {k: d.get(k) for k in ("value", "ms_per_step", "iters_mean", "iters_max", "max_relres", "solve_only_pairs_per_s")})
PY
