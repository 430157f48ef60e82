#!/usr/bin/env python3
"""Probe: PCG iterations against the Jacobi weight omega_s on rasters the bench does not cover (4-neighbour, NODATA
holes, strongly heterogeneous conductances, avg_resistances), 16 pairs each."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import circuitscape_jl_amd  # noqa
from circuitscape_jl_amd import lib as L
L.load(os.environ.get("CSGPU_LIB"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.default_rng(11)
base = rng.standard_normal((N, N))
cases = {
    "8-neigh": (np.exp(base), {}),
    "4-neigh": (np.exp(base), {"four_neighbors": True}),
    "4-neigh homogeneous": (np.ones((N, N)), {"four_neighbors": True}),
    "8-neigh homogeneous": (np.ones((N, N)), {}),
    "8-neigh exp(3 randn)": (np.exp(3 * base), {}),
    "8-neigh avg_resistances": (np.exp(base), {"avg_resistances": True}),
    "8-neigh 15% holes": (np.where(rng.random((N, N)) < 0.15, 0.0, np.exp(base)), {}),
}
for name, (g, kw) in cases.items():
    row = {"case": name, "N": N}
    for ws in (1.5, 1.6, 1.7, 1.8):
        with L.raster_setup(g, L.default_opts(batch=16, precond_bytes=4, omega_s=ws), **kw) as h:
            n = h.info["n"]
            ids = np.random.default_rng(5).choice(n, size=32, replace=False)
            R, _, _, st = h.solve_pairs([int(v) for v in ids[:16]], [int(v) for v in ids[16:]])
            row["ws%.1f" % ws] = [st["total_iters"] / 16.0, st["max_iters"] if "max_iters" in st else None, st["not_converged"]]
    print(json.dumps(row), flush=True)
