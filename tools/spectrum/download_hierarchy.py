#!/usr/bin/env python3
"""Set a raster up on the emulator build and pickle its hierarchy (A_l, P_l of every level), the raster and the component
labels for tools/spectrum/model.py.  usage: download_hierarchy.py N SEED FRAC [OUT.pkl]"""
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import circuitscape_jl_amd  # noqa: E402,F401
from circuitscape_jl_amd import lib as L  # noqa: E402

L.load(os.environ.get("CSGPU_LIB", os.path.join(ROOT, "tests", "emu", "libcsgpu_emu.so")))
N, seed, frac = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
out = sys.argv[4] if len(sys.argv) > 4 else "/tmp/nd_%d_%d_%g.pkl" % (N, seed, frac)
rng = np.random.default_rng(seed)
g = 1.0 / np.exp(rng.standard_normal((N, N)))
g = np.where(rng.random((N, N)) < frac, 0.0, g)
os.environ.setdefault("CSGPU_ENRICH", "0")   # the model adds its own enrichment
with L.raster_setup(g, L.default_opts(batch=8, precond_bytes=0)) as h:
    info = h.info
    print(info["n"], info["level_n"], info["level_form"])
    mats = {}
    for lvl in range(info["levels"]):
        for w in ("A", "P"):
            M = h.level_matrix(lvl, w)
            if M.shape[0] > 0:
                mats[(lvl, w)] = M
    labels, _ = h.components()
    ids = np.random.default_rng(5).choice(np.flatnonzero(labels == np.bincount(labels).argmax()), size=16, replace=False)
    its = [h.solve_pairs([int(a)], [int(b)])[3]["total_iters"] for a, b in zip(ids[:8], ids[8:])]
    print("library iterations per pair:", its)
pickle.dump({"mats": mats, "g": g, "labels": labels, "info": info, "library_iters": its}, open(out, "wb"))
print("wrote", out)
