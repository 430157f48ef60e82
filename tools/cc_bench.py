"""Device graph layer timing (scope row N4): raster -> CSR Laplacian + AMG setup, node map, connected components.
Usage (GPU box): python tools/cc_bench.py [size ...]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import circuitscape_jl_amd  # noqa: F401,E402
from circuitscape_jl_amd import lib  # noqa: E402

for N in [int(a) for a in sys.argv[1:]] or [2000, 5000]:
    for holes in (0.0, 0.3):
        rng = np.random.default_rng(1)
        g = np.exp(rng.standard_normal((N, N)))
        if holes:
            g[rng.random((N, N)) < holes] = 0.0
        t0 = time.perf_counter()
        h = lib.raster_setup(g, lib.default_opts(batch=8, precond_bytes=4))
        t1 = time.perf_counter()
        labels, nc = h.components()
        t2 = time.perf_counter()
        nm = h.raster_nodemap()
        t3 = time.perf_counter()
        info = h.info
        print(json.dumps({"size": N, "holes": holes, "n": info["n"], "nnz": info["nnz"], "components": nc,
                          "setup_wall_s": t1 - t0, "graph_build_ms": info["upload_ms"], "amg_setup_ms": info["setup_ms"],
                          "components_s": t2 - t1, "nodemap_copy_s": t3 - t2}), flush=True)
        h.close()
