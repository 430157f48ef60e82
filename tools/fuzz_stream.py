#!/usr/bin/env python3
"""Fuzzer for the streaming pair solves (pcg_stream_pairs) and the wide batches: random rasters (size, sigma, NODATA, 4- / 8-
neighbourhood), pair lists of random length with degenerate pairs and repeated nodes, gathered nodes, K in {8, 16, 32},
fp64 / fp32 hierarchy -- one handle, the same list through the batch path (CSGPU_NO_STREAM=1), the stream from the first
pair (CSGPU_STREAM=1) and the adaptive rule: resistances, gathered voltages and iteration counts must agree (fp64
hierarchy: bit for bit; fp32: 1e-9, see tests/helpers.py::check_stream_pairs), nothing may fail to converge that the batch
path converges. No direct solves: fast on the device.
usage: fuzz_stream.py NCASES [SEED]   env CSGPU_LIB (default: the emulator build), FUZZ_MIN / FUZZ_MAX (cells a side)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import circuitscape_jl_amd  # noqa: E402,F401
from circuitscape_jl_amd import lib as L  # noqa: E402

L.load(os.environ.get("CSGPU_LIB", os.path.join(ROOT, "tests", "emu", "libcsgpu_emu.so")))
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lo_, hi_ = int(os.environ.get("FUZZ_MIN", "30")), int(os.environ.get("FUZZ_MAX", "90"))
bad = 0
for case in range(ncases):
    rng = np.random.default_rng(seed0 * 7919 + case)
    R, C = int(rng.integers(lo_, hi_)), int(rng.integers(lo_, hi_))
    sigma = float(rng.choice([0.5, 1.0, 2.0, 3.0]))
    frac = float(rng.choice([0.0, 0.1, 0.3]))
    four = bool(rng.random() < 0.3)
    K = int(rng.choice([8, 16, 32]))
    pb = int(rng.choice([0, 4]))
    g = np.exp(sigma * rng.standard_normal((R, C)))
    g[rng.random((R, C)) < frac] = 0.0
    tag = dict(case=case, shape=(R, C), sigma=sigma, frac=frac, four=four, K=K, pb=pb)
    try:
        # (round 6: the three modes are options of a handle -- csgpu_opts.stream = -1 / 1 / 0 -- not environment variables
        # looked at per call; the three handles are the same hierarchy, built by deterministic kernels)
        with L.raster_setup(g, L.default_opts(batch=K, precond_bytes=pb, check_every=1, stream_min=1, fixed_k=1, stream=-1), four_neighbors=four) as h:
            lab, _ = h.components()
            big = np.flatnonzero(lab == np.bincount(lab).argmax())
            if len(big) < 8 or h.info["lattice_period"] == 0:
                continue
            npts = int(rng.integers(4, 14))
            pts = rng.choice(big, size=min(npts, len(big)), replace=False)
            npairs = int(rng.integers(K + 1, 4 * K + 3))
            src = [int(pts[rng.integers(0, len(pts))]) for _ in range(npairs)]
            dst = [int(pts[rng.integers(0, len(pts))]) for _ in range(npairs)]   # degenerate pairs (src == dst) happen
            gather = [int(v) for v in pts[:int(rng.integers(0, 4))]]
            out = {}
            Rr, Gv, _, st = h.solve_pairs(src, dst, gather=gather if gather else None)
            out["batch"] = (Rr, Gv, st)
            hpb = h.info["precond_bytes"]
        for mode, sm in (("stream", 1), ("adaptive", 0)):
            with L.raster_setup(g, L.default_opts(batch=K, precond_bytes=pb, check_every=1, stream_min=1, fixed_k=1, stream=sm), four_neighbors=four) as h:
                Rr, Gv, _, st = h.solve_pairs(src, dst, gather=gather if gather else None)
                out[mode] = (Rr, Gv, st)
        if True:
            Rb, Gb, sb = out["batch"]
            ok = sb["not_converged"] == 0
            for mode in ("stream", "adaptive"):
                Rs, Gs, ss = out[mode]
                nz = Rb != 0
                err = float(np.max(np.abs(Rb[nz] - Rs[nz]) / np.abs(Rb[nz]))) if np.any(nz) else 0.0
                gerr = float(np.max(np.abs(Gb - Gs))) if Gb is not None else 0.0
                tol = 0.0 if hpb == 8 else 1e-9
                ok = ok and ss["not_converged"] == 0 and err <= tol and gerr <= tol * 10 + (0 if tol == 0 else 1e-12) \
                    and ss["total_iters"] == sb["total_iters"] and np.array_equal(Rb == 0, Rs == 0)
                if mode == "stream":
                    ok = ok and ss["stream_slots"] > 0
                tag["err_" + mode] = err
            tag.update(hpb=hpb, npairs=npairs, iters=sb["total_iters"] / npairs, slots=out["stream"][2]["stream_slots"],
                       adaptive_slots=out["adaptive"][2]["stream_slots"], ok=bool(ok))
            print(json.dumps(tag), flush=True)
            if not ok:
                bad += 1
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(json.dumps(dict(tag, error=repr(e)[:300])), flush=True)
for k in ("CSGPU_NO_STREAM", "CSGPU_STREAM", "CSGPU_STREAM_MIN"):
    os.environ.pop(k, None)
print(json.dumps({"cases": ncases, "failed": bad}))
sys.exit(1 if bad else 0)
