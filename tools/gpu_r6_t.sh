#!/bin/bash
# Round 6: device fuzz of the enriched levels on the fused pass (tools/fuzz_enrich_fused.py)
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6t
rm -rf $OUT; mkdir -p $OUT
export CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu.so
timeout 900 python tools/fuzz_enrich_fused.py 71 150 > $OUT/fuzz_enrich_fused_71.jsonl 2> $OUT/f71.err; tail -1 $OUT/fuzz_enrich_fused_71.jsonl
timeout 900 python tools/fuzz_enrich_fused.py 72 150 > $OUT/fuzz_enrich_fused_72.jsonl 2> $OUT/f72.err; tail -1 $OUT/fuzz_enrich_fused_72.jsonl
FUZZ_MIN=300 FUZZ_MAX=1200 timeout 900 python tools/fuzz_enrich_fused.py 73 40 > $OUT/fuzz_enrich_fused_73_large.jsonl 2> $OUT/f73.err; tail -1 $OUT/fuzz_enrich_fused_73_large.jsonl
grep -h '"ok": false' $OUT/*.jsonl | head -8 | cut -c1-500
tail -2 $OUT/*.err
