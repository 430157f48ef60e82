#!/usr/bin/env python3
"""A/B of the 25-point lattice form of refined-tile levels (dia25.h) on ONE raster with 15 % random NODATA cells: CSR SpMM
levels (CSGPU_DIA25=0) against the marching kernel in its variants (CSGPU_DIA25_PF = load b / dinv one column ahead,
CSGPU_DIA25_WAVES = waves per SIMD the registers are held to). The kernel knobs are read at launch, so the variants share a
handle; every variant solves the same pairs (batches of K, batch path). One JSON line per (precision, variant).
usage: dia25_ab.py [SIZE]    env: PAIRS=64 BATCH=32 PBS=0,4 CSGPU_LIB"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import circuitscape_jl_amd  # noqa: E402,F401
from circuitscape_jl_amd import lib as L  # noqa: E402

L.load(os.environ.get("CSGPU_LIB"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
P = int(os.environ.get("PAIRS", "64"))
B = int(os.environ.get("BATCH", "32"))
rng = np.random.default_rng(11)
z = rng.standard_normal((N, N))
g = np.where(rng.random((N, N)) < 0.15, 0.0, np.exp(z))
del z
os.environ["CSGPU_NO_STREAM"] = "1"
VARIANTS = (("pf1_w3", "1", "3"), ("pf0_w3", "0", "3"), ("pf1_w1", "1", "1"), ("pf0_w1", "0", "1"), ("pf1_w3#2", "1", "3"))
for pb in [int(v) for v in os.environ.get("PBS", "0,4").split(",")]:
    ref = None
    for form in ("csr", "dia25"):
        os.environ["CSGPU_DIA25"] = "0" if form == "csr" else "1"
        with L.raster_setup(g, L.default_opts(batch=B, precond_bytes=pb)) as h:
            info = h.info
            labels, _ = h.components()
            pool = np.flatnonzero(labels == np.bincount(labels).argmax())
            pts = np.random.default_rng(5).choice(pool, size=15, replace=False)
            pairs = [(int(pts[i]), int(pts[j])) for i in range(15) for j in range(i + 1, 15)][:P]
            src, dst = [p[0] for p in pairs], [p[1] for p in pairs]
            h.solve_pairs(src[:B], dst[:B])          # warm-up (work vectors, code objects)
            for name, pf, wv in ((("csr", "1", "3"), ("csr#2", "1", "3")) if form == "csr" else VARIANTS):
                os.environ["CSGPU_DIA25_PF"] = pf
                os.environ["CSGPU_DIA25_WAVES"] = wv
                if form == "dia25":
                    h.solve_pairs(src[:B], dst[:B])  # (first launch of the variant's code objects)
                t0 = time.perf_counter()
                R, _, _, st = h.solve_pairs(src, dst)
                ms = (time.perf_counter() - t0) * 1e3
                if ref is None:
                    ref = R
                print(json.dumps({"N": N, "precond_bytes": info["precond_bytes"], "batch": B, "variant": name, "pairs": len(src),
                                  "ms_per_16_pairs": ms * 16.0 / len(src), "iters_mean": st["total_iters"] / float(len(src)),
                                  "iters_max": st["max_iters"], "not_converged": st["not_converged"],
                                  "max_rel_diff_vs_csr": float(np.max(np.abs(R - ref) / np.abs(ref))),
                                  "levels": info["levels"], "level_n": info["level_n"][:4]}), flush=True)
