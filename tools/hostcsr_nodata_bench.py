#!/usr/bin/env python3
"""The Julia host path on a raster with NODATA cells: the graph is built on the host the way the reference does
(construct_node_map / construct_graph / connected components, restated in oracle/refgraph.py -- test infrastructure, used
here only to produce the INPUT), its largest component goes to csgpu_setup as a compact CSR Laplacian (Int64 / 1-based, as
Julia stores it) with the raster cell of every node. With the coordinates the library scatters it into the bounding-box
lattice (cell space, marching kernels); CSGPU_NO_CELLSPACE_FROM_CSR=1 keeps the compact CSR kernels (A/B).
usage: hostcsr_nodata_bench.py [SIZE] [HOLE_FRACTION]"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import circuitscape_jl_amd  # noqa: E402,F401
from circuitscape_jl_amd import lib as L  # noqa: E402
from circuitscape_jl_amd import solver as ps  # noqa: E402
from oracle import refgraph as rg, refsolve as rs  # noqa: E402

L.load(os.environ.get("CSGPU_LIB"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.15
rng = np.random.default_rng(11)
g = np.where(rng.random((N, N)) < frac, 0.0, np.exp(rng.standard_normal((N, N))))
t0 = time.perf_counter()
nm = rg.construct_node_map(g, None)
W = rg.construct_graph(g, nm, False, False)
ncomp, labels = sp.csgraph.connected_components(W, directed=False)
comp = np.flatnonzero(labels == np.bincount(labels).argmax()) + 1
A = rs.regularize(sp.csr_matrix(rg.laplacian(W))[comp - 1][:, comp - 1])
row, col = ps._node_coords(nm, comp)
t_host = time.perf_counter() - t0
ids = np.random.default_rng(5).choice(len(comp), size=64, replace=False)
for pb in (0, 4):
    t0 = time.perf_counter()
    with L.setup(A, L.default_opts(batch=16, precond_bytes=pb), node_row=row, node_col=col) as h:
        t_setup = time.perf_counter() - t0
        ms, its = [], []
        for s in range(2):
            src = [int(v) for v in ids[32 * s:32 * s + 16]]
            dst = [int(v) for v in ids[32 * s + 16:32 * s + 32]]
            t1 = time.perf_counter()
            R, _, _, st = h.solve_pairs(src, dst)
            if s > 0:
                ms.append((time.perf_counter() - t1) * 1e3)
                its.append(st["total_iters"] / 16.0)
        info = h.info
        print(json.dumps({"N": N, "holes": frac, "n": info["n"], "rows": info["level_n"][0], "lattice_period": info["lattice_period"],
                          "knob_off": bool(os.environ.get("CSGPU_NO_CELLSPACE_FROM_CSR")), "precond_bytes": pb or 8,
                          "ms_per_batch16": float(np.mean(ms)), "iters_mean": float(np.mean(its)), "not_converged": st["not_converged"],
                          "setup_wall_s": t_setup, "setup_device_s": info["setup_ms"] / 1e3, "upload_s": info["upload_ms"] / 1e3,
                          "host_graph_build_s": t_host, "R0": float(R[0])}), flush=True)
