#!/usr/bin/env python3
"""Rasters with NODATA cells: cell space (lattice kernels on the full R x C raster, round 3) against the compact
numbering of round 2 (CSR kernels, MIS(2) aggregates) and against the all-valid raster of the same generator.
One JSON line per case: ms per batch of 16 pair solves, iterations, setup time, bytes held.
usage: nodata_bench.py SIZE [HOLE_FRACTION ...]   (env CSGPU_LIB: library to load)"""
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
fracs = [float(v) for v in sys.argv[2:]] or [0.15]
steps = int(os.environ.get("STEPS", "3"))
rng = np.random.default_rng(11)
base = np.exp(rng.standard_normal((N, N)))
mask_u = rng.random((N, N))


def blobs(frac):
    """NODATA in contiguous blobs (lakes / sea): low-pass filtered noise thresholded at the requested fraction"""
    k = max(4, N // 64)
    coarse = np.random.default_rng(3).standard_normal((N // k + 2, N // k + 2))
    field = np.kron(coarse, np.ones((k, k)))[:N, :N]
    return field < np.quantile(field, frac)


cases = [("all-valid", base, None)]
for f in fracs:
    cases.append(("%.0f%% random holes" % (100 * f), np.where(mask_u < f, 0.0, base), f))
    cases.append(("%.0f%% holes in blobs" % (100 * f), np.where(blobs(f), 0.0, base), f))
for name, g, f in cases:
    for mode in (("cell", "compact") if f is not None else ("lattice",)):
        if os.environ.get("MODES") and mode not in os.environ["MODES"].split(","):
            continue
        if mode == "compact":
            os.environ["CSGPU_NO_CELLSPACE"] = "1"
        else:
            os.environ.pop("CSGPU_NO_CELLSPACE", None)
        for pb in [int(v) for v in os.environ.get("PBS", "0,4").split(",")]:
            t0 = time.perf_counter()
            with L.raster_setup(g, L.default_opts(batch=16, precond_bytes=pb)) as h:
                t_setup = time.perf_counter() - t0
                info = h.info
                if f is None:
                    pool = np.arange(N * N)
                else:
                    labels, _ = h.components()
                    pool = np.flatnonzero(labels == np.bincount(labels).argmax())
                ids = np.random.default_rng(5).choice(pool, size=32 * (steps + 1), replace=False)
                its, ms, R = [], [], None
                for s in range(steps + 1):
                    src = [int(v) for v in ids[32 * s:32 * s + 16]]
                    dst = [int(v) for v in ids[32 * s + 16:32 * s + 32]]
                    t1 = time.perf_counter()
                    R, _, _, st = h.solve_pairs(src, dst)
                    if s > 0:
                        ms.append((time.perf_counter() - t1) * 1e3)
                        its.append(st["total_iters"] / 16.0)
                print(json.dumps({"case": name, "N": N, "mode": mode, "precond_bytes": pb or 8, "n": info["n"],
                                  "rows": info["level_n"][0], "levels": info["levels"], "lattice_period": info["lattice_period"],
                                  "ms_per_batch16": float(np.mean(ms)), "iters_mean": float(np.mean(its)),
                                  "iters_max": st["max_iters"], "not_converged": st["not_converged"],
                                  "setup_wall_s": t_setup, "setup_device_s": info["setup_ms"] / 1e3,
                                  "device_GB": info["device_bytes"] / 1e9, "R0": float(R[0])}), flush=True)
os.environ.pop("CSGPU_NO_CELLSPACE", None)
