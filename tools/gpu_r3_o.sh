#!/bin/bash
# Round 3, final GPU call: A/B of the Dirichlet correction in one process (tools/dirichlet_ab.py), the whole GPU suite on
# the final build, the driver's bench command.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3o
rm -rf $OUT; mkdir -p $OUT
timeout 90 python tools/dirichlet_ab.py 2000 16 > $OUT/dirichlet_ab_2000.jsonl 2> $OUT/dirichlet_ab.err; cat $OUT/dirichlet_ab_2000.jsonl | cut -c1-300
timeout 330 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("bench", {k: d.get(k) for k in ("value", "ms_per_step", "value_mixed", "iters_mean", "max_relres")}, d["roofline"]["frac"], d.get("parity", {}).get("max_rel_err_vs_oracle"))
except Exception as e:
    print("bench line missing", e)
PY
