#!/bin/bash
# Round 4, GPU call H: where does a 10000^2 raster with 15 % NODATA spend its time at K = 16 and K = 32 (fp64)? rocprofv3 kernel
# stats of one 96-pair call per batch width; GPU twin of the polygon test again.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4h
rm -rf $OUT; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "polygon_rasters_on_the_lattice" > $OUT/pytest_poly.log 2>&1; grep -v "^csgpu" $OUT/pytest_poly.log | tail -4
cd /tmp && export TMPDIR=/tmp
for B in 16 32; do
  MODES=batch PBS=0 BATCHES=$B PAIRS=96 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw$B -o nd -- python $GRAFT_REPO_ROOT/tools/stream_bench.py 10000 holes15 > $OUT/nodata_k$B.jsonl 2> $OUT/nodata_k$B.err
  find $OUT/raw$B -name "*kernel_stats.csv" -exec cp {} $OUT/nodata_kernel_stats_k$B.csv \;
  rm -rf $OUT/raw$B
  cut -c1-200 $OUT/nodata_k$B.jsonl
  python - $OUT/nodata_kernel_stats_k$B.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms %.1f" % (tot / 1e6))
for r in rows[:14]:
    print("  %-92s calls=%5s total_ms=%8.1f avg_us=%9.1f pct=%4.1f" % (r["Name"][:92], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
done
