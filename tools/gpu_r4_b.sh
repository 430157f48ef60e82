#!/bin/bash
# Round 4, GPU call B: batches of 32 columns -- GPU twin of the K = 32 test, then A/B of --batch 16 / --batch 32 on one box
# (fp64 path = value, mixed beside it), short runs.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4b
rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batches_of_32" > $OUT/pytest_k32.log 2>&1; tail -5 $OUT/pytest_k32.log
for B in 16 32; do
  timeout 400 python bench.py --batch $B --steps 8 --warmup 2 --cpu-sample 0 --host-csr 0 --extra-legs 0 > $OUT/bench_k$B.json 2> $OUT/bench_k$B.err
  python - $OUT/bench_k$B.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    m = d.get("mixed_path", {})
    print("batch", d["config"]["batch"], "fp64: value %.2f ms_per_step %.1f iters %.2f/%d roof %.3f (%.2f ms) | mixed: value %.2f ms %.1f iters %.2f roof %.3f | setup %.3f"
          % (d["value"], d["ms_per_step"], d["iters_mean"], d["iters_max"], d["roofline"]["frac"], d["roofline"]["avg_ms"],
             m.get("value", 0), m.get("ms_per_step", 0), m.get("iters_mean", 0), m.get("roofline", {}).get("frac", 0), d["setup_s"]))
except Exception as e:
    print("bench line missing", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
