#!/bin/bash
# Round 5: where do the remaining NODATA iterations sit? sensitivity to the coarse levels' smoothing (nu_coarse) and to the
# prolongator / smoother weights, enrichment on (tau 0.06), 10000^2 seed 2468 and 3000^2.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5e
rm -rf $OUT; mkdir -p $OUT
for o in "" "nu_coarse=3" "nu_coarse=4" "nu_coarse=6"; do
  OPTS=$o PAIRS=32 timeout 300 python tools/nodata_iters.py 10000 2468 0.06 >> $OUT/sens_10000.jsonl 2>> $OUT/err.log
done
CSGPU_TAIL_ROWS=0 OPTS="nu_coarse=4" PAIRS=32 timeout 300 python tools/nodata_iters.py 10000 2468 0.06 >> $OUT/sens_10000.jsonl 2>> $OUT/err.log
FRAC=0 PAIRS=32 OPTS="nu_coarse=4" timeout 300 python tools/nodata_iters.py 10000 0 0 >> $OUT/sens_10000.jsonl 2>> $OUT/err.log
python - <<'PY'
import json,os
for ln in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5e/sens_10000.jsonl"):
    d=json.loads(ln); print("  frac %.2f tau %.2f opts %s iters %.2f/%d ms16 %.1f" % (d["frac"],d["tau"],d["opts"],d["iters_mean"],d["iters_max"],d["ms_per_16_pairs"]))
PY
tail -n 3 $OUT/err.log
