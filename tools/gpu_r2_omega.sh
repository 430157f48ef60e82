#!/bin/bash
# omega_p / omega_s re-tuned after the near-kernel fix.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2omega2
rm -rf $OUT; mkdir -p $OUT
run() {  # size steps omega_p omega_s
  tag=n$1_p$3_s$4
  timeout 600 python bench.py --size $1 --steps $2 --warmup 1 --cpu-sample 0 --compare-steps 0 --host-csr 0 --opt omega_p=$3 --opt omega_s=$4 > $OUT/$tag.json 2> $OUT/$tag.err
  python - $OUT/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], {k: d.get(k) for k in ("value", "ms_per_step", "iters_mean", "iters_max", "max_relres")})
PY
}
for s in 1.5 1.6 1.7 1.8; do run 1000 20 1.6 $s; done
for s in 1.5 1.6 1.7 1.8; do run 3000 10 1.6 $s; done
for s in 1.5 1.6 1.7 1.8; do run 5000 6 1.6 $s; done
for s in 1.6 1.7 1.8; do run 10000 4 1.6 $s; done
for p in 1.5 1.7; do run 10000 4 $p 1.7; done
