#!/bin/bash
# Round 6, GPU call I: parameter sweep on the NODATA raster at 10000^2 (fp64, K = 32, mask 2468 and 1) with the options that
# used to be compile-time / environment choices: smoother and prolongator weights, enrichment threshold / power steps,
# sweeps on level 1, strength filter.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6i
rm -rf $OUT; mkdir -p $OUT
for O in "" "omega_s=1.6" "omega_s=1.8" "omega_p=1.5" "omega_p=1.7" "enrich_steps=12" "nu_l1=3" "nu_deep=4" "tile_theta=0.06" "coarse_smoother=2"; do
  PB=0 OPTS=$O timeout 600 python tools/nodata_iters.py 10000 2468,1 0.06 >> $OUT/sweep.jsonl 2>> $OUT/err
done
for T in 0.05 0.08; do
  PB=0 timeout 600 python tools/nodata_iters.py 10000 2468,1 $T >> $OUT/sweep.jsonl 2>> $OUT/err
done
python - <<'PY'
import json, os
for ln in open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r6i/sweep.jsonl"):
    d = json.loads(ln); print("seed %5d tau %.2f %-22s iters %.2f/%d ms16 %.1f setup %.0f" % (d["mask_seed"], d["tau"], d["opts"], d["iters_mean"], d["iters_max"], d["ms_per_16_pairs"], d["setup_device_ms"]))
PY
