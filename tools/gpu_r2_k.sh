#!/bin/bash
# Round 2, call K: DIA_CG keeps the two loaded vectors raw until the column is stored (loads overlap the previous column's
# product) -- A/B against the previous build (libcsgpu_head.so).
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2k
rm -rf $OUT; mkdir -p $OUT
B="python bench.py --compare-steps 0 --cpu-sample 0 --steps 5"
timeout 200 $B > $OUT/rawloads.json 2> $OUT/rawloads.err
CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu_head.so timeout 200 $B > $OUT/head.json 2> $OUT/head.err
timeout 200 $B --precond same > $OUT/rawloads_fp64.json 2> $OUT/rawloads_fp64.err
CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu_head.so timeout 200 $B --precond same > $OUT/head_fp64.json 2> $OUT/head_fp64.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r2k/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "value %.2f ms/step %.1f cg_prod_ms %.3f frac %.3f iters %.2f relres %.2e setup %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_ms"], d["roofline"]["frac"], d["iters_mean"], d["max_relres"], d["setup_s"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e, open(f[:-5]+".err").read()[-500:])
PY
