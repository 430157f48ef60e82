#!/usr/bin/env python3
"""Probe: iteration counts / residuals of the single-launch coarse tail against the launch-per-product V-cycle."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import circuitscape_jl_amd  # noqa
from circuitscape_jl_amd import lib as L
L.load(os.environ.get("CSGPU_LIB"))
for shape in ((700, 500), (400, 300)):
    rng = np.random.default_rng(shape[0])
    g = np.exp(rng.standard_normal(shape)); g[rng.random(shape) < 0.1] = 0.0
    n = int((g > 0).sum())
    for batch in (1, 2):
        ids = rng.choice(n, size=2 * batch, replace=False)
        src, dst = [int(v) for v in ids[:batch]], [int(v) for v in ids[batch:]]
        for pb in (0, 4):
            for graph in (0, -1):
                out = []
                for rows in (None, "0"):
                    if rows is None: os.environ.pop("CSGPU_TAIL_ROWS", None)
                    else: os.environ["CSGPU_TAIL_ROWS"] = rows
                    with L.raster_setup(g, L.default_opts(batch=batch, precond_bytes=pb, use_graph=graph)) as h:
                        R, _, _, st = h.solve_pairs(src, dst)
                        out.append({k: st[k] for k in ("total_iters", "max_relres", "polished", "graph_launches") if k in st})
                        out[-1]["levels"] = h.info["levels"]; out[-1]["R0"] = float(R[0])
                print(json.dumps({"shape": shape, "batch": batch, "pb": pb, "graph": graph, "tail": out[0], "classic": out[1]}))
