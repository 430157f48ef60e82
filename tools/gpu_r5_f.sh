#!/bin/bash
# Round 5: enrichment set-up after its optimisation (rocprofv3 kernel stats), device twin of the enrichment test, the
# NODATA / dia25 / stream tests on the device.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5f
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CSGPU_VERBOSE=1 PAIRS=32 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o nd -- python $GRAFT_REPO_ROOT/tools/nodata_iters.py 10000 2468 0.06 > $OUT/prof.jsonl 2> $OUT/prof.err
cd $GRAFT_REPO_ROOT
find $OUT/raw -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/raw
grep -E "enrich|Name" $OUT/kernel_stats.csv | cut -c1-90,250-330
cat $OUT/prof.jsonl | cut -c1-400
timeout 900 python -m pytest tests -m gpu -q -x -k "enrichment or nodata or 25_point or stream or cellspace or polygon or heterogeneous" > $OUT/pytest_subset.log 2>&1; tail -5 $OUT/pytest_subset.log
