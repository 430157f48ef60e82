#!/usr/bin/env python3
"""Fuzzer for rasters with short-circuit polygons: random small rasters (size, heterogeneity, NODATA fraction, 4- / 8-
neighbourhood, conductance / resistance averaging) with random polygons (rectangles, ragged blobs, lines, single cells,
polygons on the raster's edge, polygons full of NODATA, ids used twice, overlaps) through csgpu_raster_setup_poly -- the
lattice path where the shape rules admit it, the merged CSR graph otherwise -- against a DIRECT solve of the merged matrix
(downloaded from the forced merged-graph handle, whose construction is pinned on the reference's goldens): resistances and
gathered focal voltages between random nodes (polygon nodes among them), fp64 and fp32 hierarchy, batch 1..8.
usage: fuzz_polygons.py NCASES [SEED]   env CSGPU_LIB (default: the emulator build)"""
import json
import os
import sys

import numpy as np
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import circuitscape_jl_amd  # noqa: E402,F401
from circuitscape_jl_amd import lib as L  # noqa: E402

L.load(os.environ.get("CSGPU_LIB", os.path.join(ROOT, "tests", "emu", "libcsgpu_emu.so")))
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bad = 0
stats = {"lattice": 0, "csr": 0}
only = [int(v) for v in os.environ.get("FUZZ_ONLY", "").split(",") if v]     # (re-run single cases of a seed)
for case in range(ncases):
    if only and case not in only:
        continue
    rng = np.random.default_rng(seed0 * 100003 + case)
    lo_, hi_ = int(os.environ.get("FUZZ_MIN", "12")), int(os.environ.get("FUZZ_MAX", "46"))
    R, C = int(rng.integers(lo_, hi_)), int(rng.integers(lo_, hi_))
    sigma = float(rng.choice([0.0, 1.0, 2.0]))
    g = np.exp(sigma * rng.standard_normal((R, C)))
    g[rng.random((R, C)) < float(rng.choice([0.0, 0.05, 0.2]))] = 0.0
    poly = np.zeros((R, C), dtype=np.int32)
    npoly = int(rng.integers(1, 7))
    for k in range(1, npoly + 1):
        kind = rng.choice(["rect", "blob", "line", "cell", "edge", "nodata", "twice"])
        h, w = int(rng.integers(2, max(3, R // 3))), int(rng.integers(2, max(3, C // 3)))
        i, j = int(rng.integers(0, R - h)), int(rng.integers(0, C - w))
        if kind == "line":
            h = 1
        if kind == "cell":
            h = w = 1
        if kind == "edge":
            i = 0 if rng.random() < 0.5 else R - h
        poly[i:i + h, j:j + w] = k
        if kind == "blob":
            poly[i:i + h, j:j + w][rng.random((h, w)) < 0.25] = 0
        if kind == "nodata":
            g[i:i + h, j:j + w][rng.random((h, w)) < 0.6] = 0.0
        if kind == "twice":
            i2, j2 = int(rng.integers(0, R - 2)), int(rng.integers(0, C - 2))
            poly[i2:i2 + 2, j2:j2 + 2] = k
    if not np.any(g > 0):
        continue
    four, avg = bool(rng.random() < 0.3), bool(rng.random() < 0.3)
    B = int(rng.choice([1, 2, 4, 8]))
    pb = int(rng.choice([0, 4]))
    tag = dict(case=case, shape=(R, C), sigma=sigma, npoly=npoly, four=four, avg_res=avg, batch=B, pb=pb)
    try:
        os.environ["CSGPU_NO_POLY_LATTICE"] = "1"
        with L.raster_setup(g, L.default_opts(batch=B), four_neighbors=four, avg_resistances=avg, polymap=poly) as h:
            A = h.level_matrix(0, "A").astype(np.float64)
            nm_ref = h.raster_nodemap()
            lab, _ = h.components()
        os.environ.pop("CSGPU_NO_POLY_LATTICE")
        big = np.flatnonzero(lab == np.bincount(lab).argmax())
        if len(big) < 4:
            continue
        nodes = rng.choice(big, size=min(len(big), 6), replace=False)
        pn = np.unique(nm_ref[poly > 0])
        pn = np.intersect1d(pn[pn > 0] - 1, big)
        if len(pn):
            nodes[0] = pn[0]
            if len(pn) > 1:
                nodes[1] = pn[-1]
        nodes = np.unique(nodes)
        if len(nodes) < 2:
            continue
        src = [int(nodes[i % len(nodes)]) for i in range(5)]
        dst = [int(nodes[(i + 1 + i // len(nodes)) % len(nodes)]) for i in range(5)]
        keep = [k for k in range(5) if src[k] != dst[k]]
        src, dst = [src[k] for k in keep], [dst[k] for k in keep]
        gather = [int(v) for v in nodes[:3]]
        free = big[1:]
        lu = spla.splu(A[free][:, free].tocsc())
        pos = {int(v): k for k, v in enumerate(free)}
        Rd = np.zeros(len(src))
        Gd = np.zeros((len(src), len(gather)))
        for k, (a, b) in enumerate(zip(src, dst)):
            rhs = np.zeros(len(free))
            if a in pos:
                rhs[pos[a]] -= 1.0
            if b in pos:
                rhs[pos[b]] += 1.0
            x = np.zeros(A.shape[0])
            x[free] = lu.solve(rhs)
            Rd[k] = x[b] - x[a]
            Gd[k] = x[gather] - x[a]
        with L.raster_setup(g, L.default_opts(batch=B, precond_bytes=pb, rtol=1e-10, atol=0.0, criterion=1, itmax=3000),
                            four_neighbors=four, avg_resistances=avg, polymap=poly) as h:
            path = "lattice" if h.info["lattice_period"] > 0 else "csr"
            stats[path] += 1
            assert np.array_equal(h.raster_nodemap(), nm_ref), "node map differs from the merged path's"
            assert h.info["n"] == A.shape[0]
            Rl, Gl, _, st = h.solve_pairs(src, dst, gather=gather)
            err = float(np.max(np.abs(Rl - Rd) / np.abs(Rd)))
            gerr = float(np.max(np.abs(Gl - Gd)) / max(1.0, float(np.max(np.abs(Gd)))))
            ok = st["not_converged"] == 0 and err < 1e-7 and gerr < 1e-7
            print(json.dumps(dict(tag, path=path, err=err, gerr=gerr, iters=st["total_iters"] / len(src),
                                  not_converged=st["not_converged"], ok=bool(ok))), flush=True)
            if not ok:
                bad += 1
    except Exception as e:  # noqa: BLE001
        os.environ.pop("CSGPU_NO_POLY_LATTICE", None)
        bad += 1
        print(json.dumps(dict(tag, error=repr(e))), flush=True)
print(json.dumps({"cases": ncases, "failed": bad, "paths": stats}))
sys.exit(1 if bad else 0)
