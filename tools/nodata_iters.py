#!/usr/bin/env python3
"""Iteration counts and time per 16 pairs on rasters with i.i.d. NODATA cells over several seeds, with
the coarse-space enrichment of csrc/enrich.h off / on at several thresholds. The raster is bench.py's (make_raster + the
NODATA mask generator of its nodata15 leg, whose own seed is 2468); focal cells: 15 cells of the giant component.
usage: nodata_iters.py SIZE SEED[,SEED...] [TAU[,TAU...]]   (TAU 0 = enrichment off)   env: PB=0|4 BATCH=32 FRAC=0.15 PAIRS=64"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import circuitscape_jl_amd  # noqa: E402,F401
from circuitscape_jl_amd import lib as L  # noqa: E402

L.load(os.environ.get("CSGPU_LIB"))
N = int(sys.argv[1])
seeds = [int(v) for v in sys.argv[2].split(",")]
taus = [float(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0,0.1").split(",")]
PB = int(os.environ.get("PB", "0"))
B = int(os.environ.get("BATCH", "32"))
FRAC = float(os.environ.get("FRAC", "0.15"))
P = int(os.environ.get("PAIRS", "64"))
EXTRA = {k: (float(v) if "." in v else int(v)) for k, v in (kv.split("=") for kv in os.environ.get("OPTS", "").split(",") if kv)}
base = bench.make_raster(N)
for seed in seeds:
    if FRAC > 0:
        rng = np.random.default_rng(seed)
        g = np.where(rng.random(base.shape) < FRAC, 0.0, base)
    else:
        g = base
    for tau in taus:
        # (round 6: options of the handle -- csgpu_opts.enrich / .enrich_tau / .stream -- instead of environment variables)
        knobs = dict(enrich=0 if tau > 0 else -1, stream=-1)
        if tau > 0:
            knobs["enrich_tau"] = tau
        knobs.update(EXTRA)
        t0 = time.perf_counter()
        with L.raster_setup(g, L.default_opts(batch=B, precond_bytes=PB, **knobs)) as h:
            t_setup = time.perf_counter() - t0
            info = h.info
            if FRAC > 0:
                labels, _ = h.components()
                pool = np.flatnonzero(labels == np.bincount(labels).argmax())
                del labels
            else:
                pool = np.arange(N * N)
            pts = np.random.default_rng(bench.NODATA_PTS_SEED).choice(pool, size=15, replace=False)
            pairs = bench.lexicographic_pairs(pts)[:P]
            src, dst = [p[0] for p in pairs], [p[1] for p in pairs]
            h.solve_pairs(src[:B], dst[:B])
            t0 = time.perf_counter()
            R, _, _, st = h.solve_pairs(src, dst)
            ms = (time.perf_counter() - t0) * 1e3
            print(json.dumps({"N": N, "frac": FRAC, "mask_seed": seed, "tau": tau, "opts": EXTRA, "precond_bytes": info["precond_bytes"], "batch": B,
                              "pairs": len(src), "iters_mean": st["total_iters"] / float(len(src)), "iters_max": st["max_iters"],
                              "ms_per_16_pairs": ms * 16.0 / len(src), "setup_device_ms": info["setup_ms"], "setup_wall_s": t_setup,
                              "not_converged": st["not_converged"], "max_relres": st["max_relres"],
                              "device_bytes": info["device_bytes"], "R0": float(R[0]),
                              "fused_restrict_solves": h.info["fused_restrict_solves"],
                              "enrich_fused": 0 if os.environ.get("CSGPU_NO_ENRICH_FUSED") else 1}), flush=True)
