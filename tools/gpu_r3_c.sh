#!/bin/bash
# Round 3, call C: cell space after the orphan fix (tiles on every level, no coupled row without weight), rasters above
# 2^31 stored entries, the whole GPU suite.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3c
rm -rf $OUT; mkdir -p $OUT
timeout 300 python tools/big_raster.py 21000 4 > $OUT/big_21000.jsonl 2> $OUT/big_21000.err; cut -c1-600 $OUT/big_21000.jsonl; tail -2 $OUT/big_21000.err
timeout 300 python tools/big_raster.py 21000 4 0.1 >> $OUT/big_21000.jsonl 2>> $OUT/big_21000.err; tail -1 $OUT/big_21000.jsonl | cut -c1-600; tail -2 $OUT/big_21000.err
timeout 300 python tools/nodata_bench.py 3000 0.15 0.4 > $OUT/nodata_3000.jsonl 2> $OUT/nodata_3000.err; cut -c1-330 $OUT/nodata_3000.jsonl
STEPS=2 timeout 420 python tools/nodata_bench.py 10000 0.15 > $OUT/nodata_10000.jsonl 2> $OUT/nodata_10000.err; cut -c1-330 $OUT/nodata_10000.jsonl
timeout 900 python -m pytest tests -m gpu -q -x --durations=8 > $OUT/pytest_gpu.log 2>&1; tail -14 $OUT/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
MODES=cell PBS=0 STEPS=2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o nodata -- python $GRAFT_REPO_ROOT/tools/nodata_bench.py 10000 0.15 > $OUT/prof_nodata.jsonl 2> $OUT/prof_nodata.err
cd $GRAFT_REPO_ROOT
find $OUT/raw -name "*kernel_stats.csv" -exec cp {} $OUT/nodata_kernel_stats.csv \;
rm -rf $OUT/raw
head -14 $OUT/nodata_kernel_stats.csv | cut -c1-200
