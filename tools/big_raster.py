#!/usr/bin/env python3
"""A raster above 2^31 / 9 = 238 M cells (no int32 CSR form): the reference documents 437 M cells as tested
(docs/src/compute.md:3; use_64bit_indexing, src/run.jl:34). Sets up SIZE x SIZE (default 21000: 441 M cells) through the
index-free pipeline (lattice_setup.h), solves a few pairs and prints one JSON line: properties (symmetry, triangle
inequality, residual check), iterations, seconds, bytes held by the handle and the device's high-water mark.
usage: big_raster.py [SIZE] [BATCH] [HOLE_FRACTION]     (env PB: precond_bytes, default 4)"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import circuitscape_jl_amd  # noqa: E402,F401
from circuitscape_jl_amd import lib as L  # noqa: E402


def big_raster(N, holes=0.0, block=3000, seed=12345):
    """log-normal conductances (sigma 1, the bench's distribution), a `block`-periodic pattern so the host does not spend
    its time in the random generator; optional NODATA cells (i.i.d., fraction `holes`)"""
    rng = np.random.default_rng(seed)
    b = np.exp(rng.standard_normal((block, block)))
    if holes > 0:
        b[rng.random((block, block)) < holes] = 0.0
    reps = (N + block - 1) // block
    return np.ascontiguousarray(np.tile(b, (reps, reps))[:N, :N])


def used_gb():
    """device memory in use as the driver sees it (rocm-smi), GB; None when the tool is unavailable"""
    try:
        out = subprocess.run(["rocm-smi", "--showmeminfo", "vram", "--json"], capture_output=True, text=True, timeout=30).stdout
        d = json.loads(out)
        card = next(iter(d.values()))
        return int(card["VRAM Total Used Memory (B)"]) / 1e9
    except Exception:
        return None


def run(N, batch, holes, pb, lib=L):
    g = big_raster(N, holes)
    t0 = time.perf_counter()
    h = lib.raster_setup(g, lib.default_opts(batch=batch, precond_bytes=pb))
    t_setup = time.perf_counter() - t0
    info = h.info
    if holes > 0:
        nm = h.raster_nodemap()
        pool = nm[nm > 0] - 1
        cells = np.random.default_rng(67890).choice(pool[:: max(1, len(pool) // 100000)], size=3, replace=False)
    else:
        cells = np.random.default_rng(67890).choice(N * N, size=3, replace=False)
    a, b, c = [int(x) for x in cells]
    src, dst = [a, b, a, b], [b, a, c, c]
    t1 = time.perf_counter()
    R, _, _, st = h.solve_pairs(src, dst)
    t_solve = time.perf_counter() - t1
    hw = used_gb()
    out = {"N": N, "cells": N * N, "n": info["n"], "stored_entries": info["nnz"], "batch": batch, "precond_bytes": pb or 8,
           "holes": holes, "levels": info["levels"], "level_n": info["level_n"][:info["levels"]],
           "lattice_period": info["lattice_period"], "setup_wall_s": t_setup, "setup_device_s": info["setup_ms"] / 1e3,
           "solve_s": t_solve, "iters_mean": st["total_iters"] / len(src), "iters_max": st["max_iters"],
           "not_converged": st["not_converged"], "max_relres": st["max_relres"], "R": [float(v) for v in R],
           "symmetry_rel": float(abs(R[0] - R[1]) / R[0]), "triangle_slack": float(R[0] + R[3] - R[2]),
           "handle_GB": h.info["device_bytes"] / 1e9, "device_used_GB_after_solve": hw}
    h.close()
    return out


if __name__ == "__main__":
    L.load(os.environ.get("CSGPU_LIB"))
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 21000
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    holes = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    print(json.dumps(run(N, batch, holes, int(os.environ.get("PB", "4")))), flush=True)
