#!/bin/bash
# Round 2, call E: residual update with recomputed A p (A/B), full GPU test-suite, one-off full-size parity + host-CSR timing.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2e
rm -rf $OUT; mkdir -p $OUT
B="python bench.py --compare-steps 0 --cpu-sample 0 --steps 5"
timeout 200 $B > $OUT/recompute.json 2> $OUT/recompute.err
CSGPU_NO_RECOMPUTE=1 timeout 200 $B > $OUT/stored_ap.json 2> $OUT/stored_ap.err
timeout 200 $B --precond same > $OUT/recompute_fp64.json 2> $OUT/recompute_fp64.err
timeout 200 $B --batch 8 > $OUT/recompute_k8.json 2> $OUT/recompute_k8.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2e/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "value %.2f ms/step %.1f cg_prod_ms %.3f iters %.2f relres %.2e" % (d["value"], d["ms_per_step"], d["roofline"]["avg_ms"], d["iters_mean"], d["max_relres"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e, open(f[:-5]+".err").read()[-500:])
PY
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
free -g | head -2; nproc
timeout 1500 python tools/full_size_checks.py --out $OUT/parity_10000.json > $OUT/full_size.log 2>&1; tail -c 3000 $OUT/full_size.log
