#!/bin/bash
# Round 2, call B: lattice-form CG product + focal-node accumulation vs the CSR / full-solution paths.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2b
rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest_parity.log 2>&1; tail -3 $OUT/pytest_parity.log
B="python bench.py --compare-steps 0 --cpu-sample 0 --steps 5"
timeout 200 $B > $OUT/stencil_focal.json 2> $OUT/stencil_focal.err
timeout 200 $B --opt explicit_check=1 > $OUT/stencil_fullx.json 2> $OUT/stencil_fullx.err
timeout 200 $B --opt stencil=-1 > $OUT/csr_focal.json 2> $OUT/csr_focal.err
timeout 200 $B --opt stencil=-1 --opt explicit_check=1 > $OUT/csr_fullx.json 2> $OUT/csr_fullx.err
CSGPU_DIA_SEG=32 timeout 200 $B > $OUT/stencil_seg32.json 2> $OUT/stencil_seg32.err
CSGPU_DIA_SEG=128 timeout 200 $B > $OUT/stencil_seg128.json 2> $OUT/stencil_seg128.err
timeout 200 $B --batch 8 > $OUT/stencil_k8.json 2> $OUT/stencil_k8.err
timeout 200 $B --precond same > $OUT/stencil_fp64.json 2> $OUT/stencil_fp64.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2b/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "value %.2f ms/step %.1f cg_prod_ms %.3f iters %.2f relres %.2e" % (d["value"], d["ms_per_step"], d["roofline"]["avg_ms"], d["iters_mean"], d["max_relres"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e, open(f[:-5]+".err").read()[-500:])
PY
