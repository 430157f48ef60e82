#!/usr/bin/env python3
"""Fuzzer for the 25-point lattice form of coarse levels under refined tiles (csrc/dia25.h): random rasters (size, sigma,
NODATA fraction, NODATA rows / columns / blobs, 4- / 8-neighbour), fp64 / fp32 hierarchy, K = 8 / 16 / 32, coarse tail off and
the form forced onto every level with >= 36 rows. Per case: the marching product of every such level against the level's CSR
operator, and pair solves against the same handle built with CSGPU_DIA25=0 (iterations within 1, resistances 1e-7 / 1e-6: both answers satisfy the stopping rule).
usage: fuzz_dia25.py SEED NCASES    (env CSGPU_LIB: library to load, default the emulator build; FUZZ_MIN / FUZZ_MAX: size)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import circuitscape_jl_amd  # noqa: E402,F401
from circuitscape_jl_amd import lib as L  # noqa: E402

L.load(os.environ.get("CSGPU_LIB", os.path.join(ROOT, "tests", "emu", "libcsgpu_emu.so")))
seed0, ncase = int(sys.argv[1]), int(sys.argv[2])
os.environ["CSGPU_TAIL_ROWS"] = "0"
os.environ["CSGPU_NO_STREAM"] = "1"
lo_, hi_ = int(os.environ.get("FUZZ_MIN", "18")), int(os.environ.get("FUZZ_MAX", "110"))
bad = forms = 0
for case in range(ncase):
    rng = np.random.default_rng(seed0 * 1000 + case)
    R, C = int(rng.integers(lo_, hi_)), int(rng.integers(lo_, hi_))
    sigma = float(rng.choice([0.5, 1.0, 2.5, 3.5]))
    frac = float(rng.choice([0.0, 0.05, 0.15, 0.3]))
    four = bool(rng.integers(0, 2))
    pb = int(rng.choice([0, 4]))
    K = int(rng.choice([8, 16, 32]))
    g = np.exp(sigma * rng.standard_normal((R, C)))
    g[rng.random((R, C)) < frac] = 0.0
    if rng.random() < 0.3:
        g[rng.integers(0, R), :] = 0.0
    if rng.random() < 0.3:
        j = int(rng.integers(0, C))
        g[:, j] = 0.0
        g[int(rng.integers(0, R - 2)):, j][:2] = 1.0          # a NODATA line with a two-cell gap
    if rng.random() < 0.3:                                     # a NODATA blob
        i0, j0 = int(rng.integers(0, R - 5)), int(rng.integers(0, C - 5))
        g[i0:i0 + int(rng.integers(3, 12)), j0:j0 + int(rng.integers(3, 12))] = 0.0
    tag = dict(case=case, shape=(R, C), sigma=sigma, frac=frac, four=four, pb=pb, K=K)
    try:
        out = {}
        for form in ("csr", "dia25"):
            os.environ["CSGPU_DIA25"] = "0" if form == "csr" else "36"
            with L.raster_setup(g, L.default_opts(batch=K, precond_bytes=pb), four_neighbors=four) as h:
                info = h.info
                labels, _ = h.components()
                big = np.flatnonzero(labels == np.bincount(labels).argmax())
                if big.size < 2 * K:
                    raise KeyError("small")
                ids = np.random.default_rng(case).choice(big, size=2 * K, replace=False)
                Rr, _, _, st = h.solve_pairs([int(v) for v in ids[:K]], [int(v) for v in ids[K:]])
                out[form] = (Rr, st["total_iters"], st["not_converged"])
                if form == "dia25":
                    for lvl in range(1, info["levels"] - 1):
                        if info["level_n"][lvl] < 36:
                            continue
                        A = h.level_matrix(lvl, "A")
                        x = np.random.default_rng(lvl).standard_normal((A.shape[0], K))
                        y, _ = h.level_spmv(lvl, "A", x)
                        ref = A @ x.astype(y.dtype)
                        err = float(np.abs(y - ref).max() / max(np.abs(ref).max(), 1e-300))
                        if err > (1e-13 if y.dtype == np.float64 else 2e-6):
                            raise AssertionError("level %d product error %.2e" % (lvl, err))
                        forms += 1
        a, b = out["csr"], out["dia25"]
        diff = float(np.max(np.abs(a[0] - b[0]) / np.abs(a[0])))
        ok = abs(a[1] - b[1]) <= max(1, K // 8) and diff < (1e-7 if pb == 0 else 1e-6) and a[2] == b[2]
        if not ok:
            bad += 1
            print(json.dumps(dict(tag, ok=False, iters=(a[1], b[1]), diff=diff, not_converged=(a[2], b[2]))), flush=True)
    except KeyError:
        continue
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(json.dumps(dict(tag, ok=False, error=repr(e)[:300])), flush=True)
    if case and case % 20 == 0:
        print("# seed", seed0, "cases done", case, "bad", bad, "level products checked", forms, flush=True)
print(json.dumps(dict(seed=seed0, cases=ncase, bad=bad, level_products_checked=forms)))
