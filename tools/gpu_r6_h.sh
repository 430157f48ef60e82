#!/bin/bash
# Round 6, GPU call H: the 25-point kernel's register bound on the fp64 NODATA path, replicated (alternating, one box).
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6h
rm -rf $OUT; mkdir -p $OUT
for rep in 1 2 3; do
  for W in 0 2; do
    PB=0 OPTS=dia25_waves=$W timeout 600 python tools/nodata_iters.py 10000 2468 0.06 >> $OUT/dia25_waves_fp64.jsonl 2>> $OUT/err
  done
done
for rep in 1 2; do
  for W in 0 2; do
    PB=0 OPTS=dia25_waves=$W timeout 600 python tools/nodata_iters.py 10000 1 0.06 >> $OUT/dia25_waves_fp64.jsonl 2>> $OUT/err
  done
done
python - <<'PY'
import json, os
for ln in open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r6h/dia25_waves_fp64.jsonl"):
    d = json.loads(ln); print("seed", d["mask_seed"], d["opts"], "iters %.2f ms16 %.1f" % (d["iters_mean"], d["ms_per_16_pairs"]))
PY
