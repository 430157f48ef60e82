#!/bin/bash
# Round 6, GPU call G: (1) fp64 / mixed set-up at 10000^2 before / after the compile-time slot indices in lattice_ap_q_kernel and
# lattice_galerkin_kernel (same box, alternating); (2) the 25-point kernel's register bound: dia25_waves 2 against the default 3
# on the NODATA raster, fp64 and mixed (csgpu_opts.dia25_waves).
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6g
rm -rf $OUT; mkdir -p $OUT
cat > $OUT/setup_ab.py <<'PY'
import sys, os, json, time, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import circuitscape_jl_amd
from circuitscape_jl_amd import lib
import bench
lib.load(os.environ.get("CSGPU_LIB"))
g = bench.make_raster(10000)
gn = bench.nodata_raster(g)
out = {"lib": os.path.basename(os.environ.get("CSGPU_LIB", "libcsgpu.so"))}
for name, ras in (("all_valid", g), ("nodata15", gn)):
    for pb in (0, 4):
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            h = lib.raster_setup(ras, lib.default_opts(batch=32, precond_bytes=pb))
            wall = time.perf_counter() - t0
            ts.append((h.info["setup_ms"], wall * 1e3))
            if rep == 2:
                _, pairs = bench.focal_pairs(10000)
                if name == "all_valid":
                    R, _, _, st = h.solve_pairs([p[0] for p in pairs[:4]], [p[1] for p in pairs[:4]])
                    out["%s_pb%d_R" % (name, pb)] = [float(x) for x in R]
            h.close()
        out["%s_pb%d_setup_device_ms" % (name, pb)] = [round(t[0], 2) for t in ts]
        out["%s_pb%d_setup_wall_ms" % (name, pb)] = [round(t[1], 1) for t in ts]
print(json.dumps(out))
PY
for rep in 1 2; do
  CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu_before_setup_opt.so timeout 600 python $OUT/setup_ab.py >> $OUT/setup_ab.jsonl 2>> $OUT/err
  CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu.so timeout 600 python $OUT/setup_ab.py >> $OUT/setup_ab.jsonl 2>> $OUT/err
done
cut -c1-900 $OUT/setup_ab.jsonl
for PB in 0 4; do
  for W in 0 2; do
    PB=$PB OPTS=dia25_waves=$W timeout 600 python tools/nodata_iters.py 10000 2468 0.06 >> $OUT/dia25_waves.jsonl 2>> $OUT/err
  done
done
python - <<'PY'
import json, os
for ln in open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r6g/dia25_waves.jsonl"):
    d = json.loads(ln); print("pb", d["precond_bytes"], d["opts"], "iters %.2f ms16 %.1f" % (d["iters_mean"], d["ms_per_16_pairs"]))
PY
