"""Launch-latency regime: pair-solve throughput with and without hipGraph replay of the PCG iteration.
Usage: python tools/graph_bench.py [sizes...]   (run on the GPU box)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import circuitscape_jl_amd  # noqa: F401
from circuitscape_jl_amd import lib

sizes = [int(s) for s in sys.argv[1:]] or [250, 500, 1000, 2000]
rows = []
for N in sizes:
    rng = np.random.default_rng(0)
    g = np.exp(rng.standard_normal((N, N)))
    cells = rng.choice(N * N, size=33, replace=False)
    src, dst = cells[:-1].astype(np.int64), cells[1:].astype(np.int64)
    for batch in (8, 16):
        row = {"size": N, "batch": batch}
        for ug in (-1, 1):
            h = lib.raster_setup(g, lib.default_opts(batch=batch, precond_bytes=4, use_graph=ug, itmax=400))
            h.solve_pairs(src[:batch], dst[:batch])  # warm-up (and graph capture)
            t0 = time.perf_counter()
            R, _, _, st = h.solve_pairs(src, dst)
            dt = time.perf_counter() - t0
            row["graph" if ug > 0 else "direct"] = {"pairs_per_s": len(src) / dt, "iters": st["max_iters"],
                                                    "ms_per_iter": st["device_ms"] / max(1, st["max_iters"]) /
                                                    (len(src) / batch), "graph_launches": st["graph_launches"]}
            h.close()
        row["speedup"] = row["graph"]["pairs_per_s"] / row["direct"]["pairs_per_s"]
        rows.append(row)
        print(json.dumps(row), flush=True)
