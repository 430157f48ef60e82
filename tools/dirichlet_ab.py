#!/usr/bin/env python3
"""A/B of the coarsest-level correction of Dirichlet-masked solves (csrc/pcg.h, DirichletCoarse) in ONE process: one-to-all
columns (unit source at one focal cell, the other focal cells grounded) on the pair-solve handle of an N x N raster, fp32 and
fp64 hierarchy, with and without the correction (CSGPU_NO_DIRICHLET_COARSE is read at setup). Prints one JSON line per case:
iterations per column, HIP-event milliseconds of the PCG loops, worst ||Ax-b||/||b||.
usage: dirichlet_ab.py [N] [points]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import circuitscape_jl_amd  # noqa
from circuitscape_jl_amd import lib
lib.load(os.environ.get("CSGPU_LIB"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 16
g = bench.make_raster(N)
rng = np.random.default_rng(4242)
pts = rng.choice(N * N, size=npts, replace=False)
B = np.zeros((N * N, npts))
grounds = []
for c, p in enumerate(pts):
    B[p, c] = 1.0
    grounds.append([int(q) for q in pts if q != p])
for pb in (4, 0):
    for off in (False, True, False, True):
        if off:
            os.environ["CSGPU_NO_DIRICHLET_COARSE"] = "1"
        else:
            os.environ.pop("CSGPU_NO_DIRICHLET_COARSE", None)
        with lib.raster_setup(g, lib.default_opts(batch=16, precond_bytes=pb)) as h:
            h.solve_grounded(B[:, :1], grounds[:1])   # warm-up of the code path
            X, _, st = h.solve_grounded(B, grounds)
        print(json.dumps({"N": N, "columns": npts, "hierarchy": "fp32" if pb == 4 else "fp64", "correction": not off,
                          "iters_mean": st["total_iters"] / npts, "iters_max": st["max_iters"], "pcg_device_ms": st["device_ms"],
                          "max_relres": st["max_relres"], "not_converged": st["not_converged"]}), flush=True)
os.environ.pop("CSGPU_NO_DIRICHLET_COARSE", None)
