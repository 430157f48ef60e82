#!/bin/bash
# Round 2, call F: index-free transfer operators (A/B against the CSR forms), parity tests, profile.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2f
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest_parity.log 2>&1; tail -3 $OUT/pytest_parity.log
B="python bench.py --compare-steps 0 --cpu-sample 0 --steps 5"
timeout 200 $B > $OUT/lattice_q.json 2> $OUT/lattice_q.err
CSGPU_NO_LATTICE_Q=1 timeout 200 $B > $OUT/csr_q.json 2> $OUT/csr_q.err
CSGPU_RESTRICT_SEG=16 timeout 200 $B > $OUT/lattice_q_rseg16.json 2> $OUT/lattice_q_rseg16.err
CSGPU_RESTRICT_SEG=64 timeout 200 $B > $OUT/lattice_q_rseg64.json 2> $OUT/lattice_q_rseg64.err
timeout 200 $B --precond same > $OUT/lattice_q_fp64.json 2> $OUT/lattice_q_fp64.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2f/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "value %.2f ms/step %.1f cg_prod_ms %.3f frac %.3f iters %.2f relres %.2e setup %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_ms"], d["roofline"]["frac"], d["iters_mean"], d["max_relres"], d["setup_s"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e, open(f[:-5]+".err").read()[-500:])
PY
TAG=r2f bash tools/gpu_r2_prof.sh
