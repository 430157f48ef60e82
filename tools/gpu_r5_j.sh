#!/bin/bash
# Round 5: which pairs are slow at 10000^2 / 15 % NODATA and what their focal cells sit in; the trend between 3000^2 and 10000^2.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5j
rm -rf $OUT; mkdir -p $OUT
timeout 600 python tools/nodata_pairs.py 10000 2468 0.06 > $OUT/pairs_10000.txt 2> $OUT/err.log
grep -v "^{" $OUT/pairs_10000.txt | head -200
PAIRS=32 timeout 300 python tools/nodata_iters.py 6000 2468 0,0.06,0.15 > $OUT/nodata_6000.jsonl 2>> $OUT/err.log
PAIRS=32 timeout 300 python tools/nodata_iters.py 10000 2468 0.15 >> $OUT/nodata_6000.jsonl 2>> $OUT/err.log
python - <<'PY'
import json,os
for ln in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5j/nodata_6000.jsonl"):
    d=json.loads(ln); print("  N %d tau %.2f iters %.2f/%d ms16 %.1f" % (d["N"],d["tau"],d["iters_mean"],d["iters_max"],d["ms_per_16_pairs"]))
PY
tail -n 3 $OUT/err.log
