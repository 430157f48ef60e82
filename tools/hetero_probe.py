#!/usr/bin/env python3
"""Probe: PCG iterations on rasters the bench does not cover -- log-normal conductances of growing spread, 4-neighbour
connectivity, NODATA holes, averaged resistances -- 16 pairs each, bench defaults (fp32 hierarchy under fp64 CG)."""
import os, sys, json, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import circuitscape_jl_amd  # noqa
from circuitscape_jl_amd import lib as L
L.load(os.environ.get("CSGPU_LIB"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.default_rng(11)
base = rng.standard_normal((N, N))
cases = [
    ("8-neigh sigma 1", np.exp(base), {}),
    ("8-neigh sigma 2", np.exp(2 * base), {}),
    ("8-neigh sigma 3", np.exp(3 * base), {}),
    ("4-neigh sigma 1", np.exp(base), {"four_neighbors": True}),
    ("8-neigh homogeneous", np.ones((N, N)), {}),
    ("8-neigh sigma 1 avg_resistances", np.exp(base), {"avg_resistances": True}),
    ("8-neigh sigma 1, 15% holes", np.where(rng.random((N, N)) < 0.15, 0.0, np.exp(base)), {}),
]
for name, g, kw in cases:
    with L.raster_setup(g, L.default_opts(batch=16, precond_bytes=4), **kw) as h:
        n = h.info["n"]
        ids = np.random.default_rng(5).choice(n, size=32, replace=False)
        t0 = time.perf_counter()
        R, _, _, st = h.solve_pairs([int(v) for v in ids[:16]], [int(v) for v in ids[16:]])
        dt = time.perf_counter() - t0
        print(json.dumps({"case": name, "N": N, "iters_mean": st["total_iters"] / 16.0, "not_converged": st["not_converged"],
                          "max_relres": st["max_relres"], "solve_s": dt, "levels": h.info["levels"]}), flush=True)
