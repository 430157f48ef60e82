#!/bin/bash
# Round 6: after the sparse first restriction on the enriched fused path: whole device suite, the fuzzer on a new seed, NODATA 5 seeds
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6u
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $OUT/pytest_gpu.log 2>&1; tail -10 $OUT/pytest_gpu.log
export CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu.so
timeout 900 python tools/fuzz_enrich_fused.py 74 120 > $OUT/fuzz_enrich_fused_74.jsonl 2> $OUT/f74.err; tail -1 $OUT/fuzz_enrich_fused_74.jsonl
grep -h '"ok": false' $OUT/*.jsonl | head -8 | cut -c1-500
unset CSGPU_LIB
timeout 600 python tools/nodata_iters.py 10000 2468,1,2,3,4 0.06 > $OUT/nodata_10000_fused.jsonl 2> $OUT/nd.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6u/nodata*.jsonl")):
    for ln in open(f):
        d=json.loads(ln); print("  seed %5d tau %.2f iters %.2f/%d ms16 %.1f setup %.0f ms nc %d fused %d" % (d["mask_seed"],d["tau"],d["iters_mean"],d["iters_max"],d["ms_per_16_pairs"],d["setup_device_ms"],d["not_converged"],d["fused_restrict_solves"]))
PY
