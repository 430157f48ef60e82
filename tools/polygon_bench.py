#!/usr/bin/env python3
"""Rasters WITH short-circuit polygons (construct_node_map with a polymap, src/raster/pairwise.jl:276-301) at scale:
csgpu_raster_setup_poly (device graph layer, CSR kernels + MIS(2)/coordinate hierarchy) against the polygon-free raster
of the same generator (index-free lattice kernels). NPOLY rectangular polygons of random size (side 0.2 % .. 2 % of the
raster) at random positions; focal points outside the polygons. One JSON line per case: setup, ms per 16 pairs, iterations.
usage: polygon_bench.py SIZE [NPOLY]   env: PBS=0,4 BATCH=16 POLY_MAX=<largest polygon side in cells>"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import circuitscape_jl_amd  # noqa: E402,F401
from circuitscape_jl_amd import lib as L  # noqa: E402

L.load(os.environ.get("CSGPU_LIB"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 50
B = int(os.environ.get("BATCH", "16"))
rng = np.random.default_rng(21)
g = np.exp(rng.standard_normal((N, N)))
poly = np.zeros((N, N), dtype=np.int32)
shape = os.environ.get("POLY_SHAPE", "rect")       # rect | lines (1-2 cells wide: rivers, roads) | mixed
for k in range(NP):
    smax = int(os.environ.get("POLY_MAX", str(max(3, N // 50))))
    h, w = rng.integers(max(2, min(N // 500, smax - 1)), smax, size=2)
    if shape == "lines" or (shape == "mixed" and k % 2):
        wd = int(os.environ.get("POLY_WIDTH", "0")) or int(rng.integers(1, 3))
        if k % 4 < 2:
            h = wd
        else:
            w = wd
    i, j = rng.integers(0, N - h), rng.integers(0, N - w)
    poly[i:i + h, j:j + w] = k + 1          # (later polygons overwrite earlier ones where they overlap)
free = np.flatnonzero(poly.ravel() == 0)
cells = np.random.default_rng(5).choice(free, size=2 * B, replace=False)
rows, cols = cells // N, cells % N
for pb in [int(v) for v in os.environ.get("PBS", "0,4").split(",")]:
    for name, pm in (("no polygons", None), ("%d polygons (%.2f %% of the cells)" % (NP, 100.0 * np.mean(poly > 0)), poly)):
        t0 = time.perf_counter()
        with L.raster_setup(g, L.default_opts(batch=B, precond_bytes=pb), polymap=pm) as h:
            t_setup = time.perf_counter() - t0
            info = h.info
            nm = h.raster_nodemap()
            nodes = nm[rows, cols].astype(np.int64) - 1
            src, dst = [int(v) for v in nodes[:B]], [int(v) for v in nodes[B:]]
            h.solve_pairs(src, dst)
            t1 = time.perf_counter()
            R, _, _, st = h.solve_pairs(src, dst)
            ms = (time.perf_counter() - t1) * 1e3
            print(json.dumps({"case": name, "N": N, "precond_bytes": info["precond_bytes"], "batch": B, "n": info["n"],
                              "levels": info["levels"], "level_n": info["level_n"], "lattice_period": info["lattice_period"],
                              "operator_complexity": info["operator_complexity"], "setup_wall_s": t_setup,
                              "setup_device_s": info["setup_ms"] / 1e3, "ms_per_batch": ms * 16.0 / B,
                              "iters_mean": st["total_iters"] / float(B), "iters_max": st["max_iters"],
                              "not_converged": st["not_converged"], "R0": float(R[0])}), flush=True)
