#!/usr/bin/env python3
"""Heterogeneous rasters: log-normal conductances exp(sigma N(0,1)), sigma = 1, 2, 3, at SIZE x SIZE;
PCG iterations and ms per batch of 16 pairs for the fp64 and the fp32 hierarchy, and (ORACLE=1) the iteration count of the
CPU oracle -- the reference's algorithm with its symmetric Gauss-Seidel smoother -- on the first two pairs.
CSGPU_TILE_THETA=0 switches the strength-aware tiles off (A/B). One JSON line per case.
usage: hetero_bench.py [SIZE]"""
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
z = np.random.default_rng(11).standard_normal((N, N))
ids = np.random.default_rng(5).choice(N * N, size=32, replace=False)
src, dst = [int(v) for v in ids[:16]], [int(v) for v in ids[16:]]
for sigma in (1.0, 2.0, 3.0):
    g = np.exp(sigma * z)
    rec = {"N": N, "sigma": sigma, "tile_theta": os.environ.get("CSGPU_TILE_THETA", "default")}
    for pb in (0, 4):
        with L.raster_setup(g, L.default_opts(batch=16, precond_bytes=pb)) as h:
            h.solve_pairs(src, dst)
            t0 = time.perf_counter()
            R, _, _, st = h.solve_pairs(src, dst)
            tag = "fp64" if pb == 0 else "mixed"
            rec.update({tag + "_iters_mean": st["total_iters"] / 16.0, tag + "_iters_max": st["max_iters"],
                        tag + "_ms_per_batch": (time.perf_counter() - t0) * 1e3, tag + "_not_converged": st["not_converged"],
                        tag + "_setup_device_s": h.info["setup_ms"] / 1e3, tag + "_R0": float(R[0])})
    if os.environ.get("ORACLE") and sigma > 1.0:
        from oracle import refgraph as rg, refsolve as rs
        A = rs.regularize(rg.laplacian(rg.construct_graph(g, rg.construct_node_map(g, None), False, False)))
        S = rs.OracleAMG(A)
        Ro, _, o = S.solve_pairs(src[:2], dst[:2], nthreads=2)
        rec.update({"oracle_iters": [x["iters"] for x in o], "oracle_R0": float(Ro[0])})
    print(json.dumps(rec), flush=True)
