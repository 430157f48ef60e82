#!/bin/bash
# Round 6: enriched NODATA levels on the fused residual update + restriction (enrich_coarse_fix): device tests, 5 mask seeds at
# 10000^2 with the coarse-side correction off (two passes, CSGPU_NO_ENRICH_FUSED=1) and on
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6q
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "enrich or fused or nodata" > $OUT/pytest_enrich.log 2>&1; tail -8 $OUT/pytest_enrich.log
CSGPU_NO_ENRICH_FUSED=1 timeout 600 python tools/nodata_iters.py 10000 2468,1,2,3,4 0.06 > $OUT/nodata_10000_twopass.jsonl 2> $OUT/nd.err
timeout 600 python tools/nodata_iters.py 10000 2468,1,2,3,4 0.06 > $OUT/nodata_10000_fused.jsonl 2>> $OUT/nd.err
timeout 300 python tools/nodata_iters.py 3000 2468,1,2 0.06 > $OUT/nodata_3000_fused.jsonl 2>> $OUT/nd.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r6q/nodata*.jsonl")):
    print(os.path.basename(f))
    for ln in open(f):
        d=json.loads(ln); print("  seed %5d tau %.2f iters %.2f/%d ms16 %.1f setup %.0f ms nc %d fused %d relres %.2e" % (d["mask_seed"],d["tau"],d["iters_mean"],d["iters_max"],d["ms_per_16_pairs"],d["setup_device_ms"],d["not_converged"],d["fused_restrict_solves"],d["max_relres"]))
PY
tail -3 $OUT/nd.err
