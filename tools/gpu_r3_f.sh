#!/bin/bash
# Round 3, call F: level 1 in lattice form (four marching products) against the CSR level 1: GPU test + bench A/B
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3f
rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -x -k "lattice_level1 or lattice_pipeline or cellspace or bench_default_batch16" > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 400 python bench.py --steps 8 --warmup 2 --cpu-sample 0 --host-csr 0 > $OUT/bench_lattice_l1.json 2> $OUT/bench_lattice_l1.err
CSGPU_NO_LATTICE_L1=1 timeout 400 python bench.py --steps 8 --warmup 2 --cpu-sample 0 --host-csr 0 > $OUT/bench_csr_l1.json 2> $OUT/bench_csr_l1.err
python - <<'PY'
import json, os
for f in ("bench_lattice_l1", "bench_csr_l1"):
    try:
        d = json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r3f", f + ".json")).read().strip().splitlines()[-1])
        m = d["mixed_path"]
        print(f, "fp64 ms/step", round(d["ms_per_step"], 1), "iters", d["iters_mean"], "setup_dev", round(d["setup_device_s"], 3),
              "| mixed ms/step", round(m["ms_per_step"], 1), "iters", m["iters_mean"], "setup_dev", round(m["setup_device_s"], 3))
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -2 $OUT/bench_lattice_l1.err
