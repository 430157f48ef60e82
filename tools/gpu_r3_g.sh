#!/bin/bash
# Round 3, call G: strength-aware tiles on heterogeneous rasters
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3g
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x -k "heterogeneous or lattice_pipeline or cellspace" > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 300 python tools/hetero_bench.py 3000 > $OUT/hetero_3000.jsonl 2> $OUT/hetero_3000.err; cut -c1-700 $OUT/hetero_3000.jsonl
CSGPU_TILE_THETA=0 timeout 300 python tools/hetero_bench.py 3000 > $OUT/hetero_3000_off.jsonl 2> $OUT/hetero_3000_off.err; cut -c1-500 $OUT/hetero_3000_off.jsonl
timeout 400 python bench.py --steps 6 --warmup 2 --cpu-sample 0 --host-csr 0 > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json, os
d = json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r3g/bench.json")).read().strip().splitlines()[-1])
m = d["mixed_path"]
print("fp64 ms/step", round(d["ms_per_step"], 1), "iters", d["iters_mean"], "setup_dev", round(d["setup_device_s"], 3),
      "| mixed ms/step", round(m["ms_per_step"], 1), "iters", m["iters_mean"], "setup_dev", round(m["setup_device_s"], 3))
PY
