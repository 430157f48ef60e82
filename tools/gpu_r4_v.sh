#!/bin/bash
# Round 4, GPU call V: dia25 kernel variants (b / dinv prefetch, register bound) -- device twin test under both register bounds,
# then the A/B on one 10000^2 raster with 15 % NODATA (tools/dia25_ab.py).
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4v
rm -rf $OUT; mkdir -p $OUT
export CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu.so
for w in 3 1; do
  CSGPU_DIA25_WAVES=$w timeout 60 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "25_point" > $OUT/pytest_dia25_w$w.log 2>&1; tail -1 $OUT/pytest_dia25_w$w.log
done
timeout 110 python tools/dia25_ab.py 10000 > $OUT/dia25_ab_10000.jsonl 2> $OUT/dia25_ab_10000.err
python - <<PY
import json
for l in open("$OUT/dia25_ab_10000.jsonl"):
    d = json.loads(l); print("pb", d["precond_bytes"], d["variant"], "ms/16", round(d["ms_per_16_pairs"], 1), "iters", round(d["iters_mean"], 2), "diff", "%.1e" % d["max_rel_diff_vs_csr"])
PY
tail -3 $OUT/dia25_ab_10000.err
