#!/bin/bash
# Round 4, GPU call T: the 25-point lattice form of refined-tile levels (dia25.h, opt-in CSGPU_DIA25): device twin test,
# A/B on the 15 % NODATA raster at 10000^2 (K = 32, fp64 and mixed), kernel-level profile of both paths at 6000^2.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4t
rm -rf $OUT; mkdir -p $OUT
export CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu.so
timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "25_point" > $OUT/pytest_dia25.log 2>&1; tail -3 $OUT/pytest_dia25.log
for mode in off on; do
  if [ $mode = on ]; then export CSGPU_DIA25=1; else unset CSGPU_DIA25; fi
  MODES=batch BATCHES=32 PBS=0,4 PAIRS=64 timeout 150 python tools/stream_bench.py 10000 holes15 > $OUT/ab_$mode.jsonl 2> $OUT/ab_$mode.err
  python - <<PY
import json
for l in open("$OUT/ab_$mode.jsonl"):
    d = json.loads(l); print("dia25 $mode pb", d["precond_bytes"], "ms/16", round(d["ms_per_16_pairs"], 1), "iters", round(d["iters_mean"], 2), d["iters_max"])
PY
done
cd /tmp && export TMPDIR=/tmp
for mode in off on; do
  if [ $mode = on ]; then export CSGPU_DIA25=1; else unset CSGPU_DIA25; fi
  MODES=batch BATCHES=32 PBS=0 PAIRS=32 timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/prof_$mode -o p -- python $GRAFT_REPO_ROOT/tools/stream_bench.py 6000 holes15 > $OUT/prof_$mode.jsonl 2> $OUT/prof_$mode.err
  f=$(find $OUT/prof_$mode -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -14 "$f" | cut -c1-220 > $OUT/prof_${mode}_kernel_stats_head.csv && head -8 $OUT/prof_${mode}_kernel_stats_head.csv | cut -c1-160
  find $OUT/prof_$mode -type f ! -name "*kernel_stats.csv" -delete
done
