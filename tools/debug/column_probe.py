#!/usr/bin/env python3
"""Is a pair's result independent of the COLUMN it is solved in and of what its neighbour columns hold? (round 4: the
streaming path's resistances agree with the batch path's bit for bit with an fp64 hierarchy but differ by 1e-12 with an
fp32 one on the device -- not on the emulator.) One handle; the same pair in every column of a batch, then the pair in
column c of a batch of other pairs, for c = 0, 5, K-1."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import circuitscape_jl_amd  # noqa: E402,F401
from circuitscape_jl_amd import lib as L  # noqa: E402

L.load(os.environ.get("CSGPU_LIB"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
K = int(os.environ.get("BATCH", "16"))
g = np.exp(np.random.default_rng(11).standard_normal((N, N)))
pts = [int(v) for v in np.random.default_rng(5).choice(N * N, size=2 * K + 2, replace=False)]
os.environ["CSGPU_NO_STREAM"] = "1"
for pb in (4, 0):
    with L.raster_setup(g, L.default_opts(batch=K, precond_bytes=pb)) as h:
        a, b = pts[0], pts[1]
        R, _, _, st = h.solve_pairs([a] * K, [b] * K)
        out = {"precond_bytes": pb or 8, "same_pair_in_every_column_identical": bool(np.all(R == R[0])),
               "spread": float((R.max() - R.min()) / R[0]), "iters": st["total_iters"] / K}
        vals = {}
        for c in (0, 5, K - 1):
            src = [pts[2 + i] for i in range(K)]
            dst = [pts[2 + K + i] for i in range(K)]
            src[c], dst[c] = a, b
            R2, _, _, _ = h.solve_pairs(src, dst)
            vals[c] = float(R2[c])
        out["pair_among_other_pairs_by_column"] = vals
        out["identical_across_positions"] = bool(len(set(vals.values())) == 1 and list(vals.values())[0] == float(R[0]))
        out["rel_diff_vs_all_same"] = {c: abs(v - float(R[0])) / float(R[0]) for c, v in vals.items()}
        print(json.dumps(out), flush=True)
