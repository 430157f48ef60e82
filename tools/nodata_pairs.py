#!/usr/bin/env python3
"""Which pairs of a NODATA raster are the slow ones, and what do their focal cells sit in? One csgpu_solve_pairs call per pair
(K = 1: the iteration count of that pair alone), then the NODATA pattern around every focal cell.
usage: nodata_pairs.py SIZE MASK_SEED [TAU]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import circuitscape_jl_amd  # noqa: E402,F401
from circuitscape_jl_amd import lib as L  # noqa: E402

L.load(os.environ.get("CSGPU_LIB"))
N, seed = int(sys.argv[1]), int(sys.argv[2])
tau = float(sys.argv[3]) if len(sys.argv) > 3 else 0.06
os.environ["CSGPU_ENRICH_TAU"] = str(tau)
os.environ["CSGPU_ENRICH"] = "1" if tau > 0 else "0"
base = bench.make_raster(N)
g = np.where(np.random.default_rng(seed).random(base.shape) < 0.15, 0.0, base)
with L.raster_setup(g, L.default_opts(batch=int(os.environ.get("BATCH", "1")))) as h:
    labels, _ = h.components()
    pool = np.flatnonzero(labels == np.bincount(labels).argmax())
    pts = np.random.default_rng(bench.NODATA_PTS_SEED).choice(pool, size=15, replace=False)
    nm, _, _ = h.nodemap() if hasattr(h, "nodemap") else (None, None, None)
    per = {}
    for a in range(15):
        b = (a + 7) % 15
        R, _, _, st = h.solve_pairs([int(pts[a])], [int(pts[b])])
        per[(a, b)] = st["total_iters"]
        print(json.dumps({"src": a, "dst": b, "iters": st["total_iters"]}), flush=True)
# node id -> cell: compact numbering = rank among the valid cells in column-major order
valid = (g.T.ravel() > 0)
n2c = np.flatnonzero(valid)
for k in range(15):
    cell = n2c[pts[k]]
    r, c = int(cell % N), int(cell // N)
    its = [v for (a, b), v in per.items() if a == k or b == k]
    print("point %d node %d cell (%d, %d) iterations of its pairs %s" % (k, pts[k], r, c, its))
    for rr in range(max(0, r - 4), min(N, r + 5)):
        print("    " + "".join(("X" if (rr == r and cc == c) else ("." if g[rr, cc] == 0 else "o")) for cc in range(max(0, c - 6), min(N, c + 7))))
