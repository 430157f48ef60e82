#!/bin/bash
# Round 3, call B: cell space (NODATA rasters on the lattice kernels): GPU tests, ms per batch / iterations against the
# compact numbering at 3000^2 and 10000^2, rocprofv3 kernel stats of the 10000^2 / 15 % holes run.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3b
rm -rf $OUT; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x -k "cellspace or nodata or region or omniscape or device_graph" > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 900 python tools/nodata_bench.py 3000 0.15 0.4 > $OUT/nodata_3000.jsonl 2> $OUT/nodata_3000.err; cat $OUT/nodata_3000.jsonl | cut -c1-330
STEPS=2 timeout 1500 python tools/nodata_bench.py 10000 0.15 > $OUT/nodata_10000.jsonl 2> $OUT/nodata_10000.err; cat $OUT/nodata_10000.jsonl | cut -c1-330
cd /tmp && export TMPDIR=/tmp
MODES=cell PBS=0 STEPS=2 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o nodata -- python $GRAFT_REPO_ROOT/tools/nodata_bench.py 10000 0.15 > $OUT/prof_nodata.jsonl 2> $OUT/prof_nodata.err
cd $GRAFT_REPO_ROOT
find $OUT/raw -name "*kernel_stats.csv" -exec cp {} $OUT/nodata_kernel_stats.csv \;
rm -rf $OUT/raw
head -16 $OUT/nodata_kernel_stats.csv | cut -c1-200
