#!/bin/bash
# hipGraph replay: small-raster throughput comparison. Outputs -> gpurun_out/graph/
OUT=$GRAFT_REPO_ROOT/gpurun_out/graph
mkdir -p $OUT
timeout 400 python tools/graph_bench.py 250 500 1000 2000 > $OUT/graph_bench.jsonl 2> $OUT/graph_bench.err; cat $OUT/graph_bench.jsonl; tail -3 $OUT/graph_bench.err
