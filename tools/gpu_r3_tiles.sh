#!/bin/bash
# Round 3: tile-shape knobs of the marching kernels re-measured on the all-fp64 path (CSGPU_DIA_SEG raster columns per
# tile, CSGPU_RESTRICT_SEG coarse columns per restriction tile); 4 steps each, one box.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3tiles
rm -rf $OUT; mkdir -p $OUT
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 4 --warmup 1 --cpu-sample 0 --host-csr 0 --compare-steps 4 > $OUT/$tag.json 2> $OUT/$tag.err
  python - $OUT/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); m = d["mixed_path"]
print(sys.argv[2], "fp64", round(d["ms_per_step"], 1), "cg", round(d["roofline"]["avg_ms"], 3), "| mixed", round(m["ms_per_step"], 1), "cg", round(m["roofline"]["avg_ms"], 3))
PY
}
run base X=1
run dia48 CSGPU_DIA_SEG=48
run dia64 CSGPU_DIA_SEG=64
run dia24 CSGPU_DIA_SEG=24
run rs16 CSGPU_RESTRICT_SEG=16
run rs64 CSGPU_RESTRICT_SEG=64
run base2 X=1
