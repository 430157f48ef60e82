#!/bin/bash
# Round 4, GPU call G: the whole GPU suite on the current build, smoke(), polygon bench (final rule) at 5000^2.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4g
rm -rf $OUT; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -6 $OUT/pytest_gpu.log
timeout 120 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
for CASE in "50 100 rect" "400 20 rect" "2000 6 rect"; do
  set -- $CASE
  POLY_MAX=$2 POLY_SHAPE=$3 PBS=0,4 timeout 200 python tools/polygon_bench.py 5000 $1 2> /dev/null >> $OUT/polygons_5000_final.jsonl
done
python - $OUT/polygons_5000_final.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print("  %-40s pb%d lat %d setup %.3fs ms/batch %.1f iters %.2f/%d" % (d["case"][:40], d["precond_bytes"], d["lattice_period"], d["setup_wall_s"], d["ms_per_batch"], d["iters_mean"], d["iters_max"]))
PY
