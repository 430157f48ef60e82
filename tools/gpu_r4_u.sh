#!/bin/bash
# Round 4, GPU call U: 25-point lattice form on by default -- the NODATA / heterogeneity / polygon / streaming GPU tests, A/B of
# its two kernels (ring: x window read from LDS; window: x window in registers) at 10000^2 15 % NODATA K = 32, kernel-level
# profiles (CSV) of the CSR path and the window kernel at 6000^2.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4u
rm -rf $OUT; mkdir -p $OUT
export CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu.so
timeout 170 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "nodata or cellspace or heterogeneous or streaming or polygon or 25_point or region_pairs_of_single" > $OUT/pytest_subset.log 2>&1; tail -3 $OUT/pytest_subset.log
for mode in ring window; do
  export CSGPU_DIA25_KERNEL=$mode
  MODES=batch BATCHES=32 PBS=0,4 PAIRS=64 timeout 100 python tools/stream_bench.py 10000 holes15 > $OUT/ab_$mode.jsonl 2> $OUT/ab_$mode.err
  python - <<PY
import json
for l in open("$OUT/ab_$mode.jsonl"):
    d = json.loads(l); print("dia25 $mode pb", d["precond_bytes"], "ms/16", round(d["ms_per_16_pairs"], 1), "iters", round(d["iters_mean"], 2), d["iters_max"])
PY
done
unset CSGPU_DIA25_KERNEL
cd /tmp && export TMPDIR=/tmp
for mode in csr window; do
  if [ $mode = csr ]; then export CSGPU_DIA25=0; else unset CSGPU_DIA25; fi
  MODES=batch BATCHES=32 PBS=0 PAIRS=32 timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$mode -o p -- python $GRAFT_REPO_ROOT/tools/stream_bench.py 6000 holes15 > $OUT/prof_$mode.jsonl 2> $OUT/prof_$mode.err
  f=$(find $OUT/prof_$mode -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then head -16 "$f" | cut -c1-260 > $OUT/prof_${mode}_kernel_stats_head.csv; head -9 $OUT/prof_${mode}_kernel_stats_head.csv | cut -c1-170; fi
  rm -rf $OUT/prof_$mode
done
