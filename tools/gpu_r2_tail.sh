#!/bin/bash
# Coarse tail in one launch (csrc/tail.h): parity test + threshold sweep at 500^2..2000^2 and at the headline size.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2tail
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x -k "coarse_tail or golden or solve_paths or two_product" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
run() {  # size rows steps
  CSGPU_TAIL_ROWS=$2 timeout 600 python bench.py --size $1 --steps $3 --warmup 2 --cpu-sample 0 --compare-steps 0 --host-csr 0 > $OUT/s$1_r$2.json 2> $OUT/s$1_r$2.err
  python - $OUT/s$1_r$2.json $1 $2 <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("size", sys.argv[2], "tail_rows", sys.argv[3], {k: d.get(k) for k in ("value", "ms_per_step", "iters_mean", "solve_only_pairs_per_s", "max_relres")})
PY
}
for s in 250 500 1000 2000; do for r in 0 2048 4096 16384; do run $s $r 20; done; done
for r in 0 4096 0 4096; do run 10000 $r 5; done
