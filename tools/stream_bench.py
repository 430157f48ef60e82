#!/usr/bin/env python3
"""Streaming pair solves (pcg_stream_pairs) against the batch path on ONE handle per case: one csgpu_solve_pairs call
with P pairs (the lexicographic pair list over 15 focal points of the giant component), modes batch (CSGPU_NO_STREAM=1),
stream (CSGPU_STREAM=1: from the first pair on) and adaptive (default: the first batch decides).
One JSON line per (case, precision, batch width, mode): ms per 16 pairs, K-wide iterations, column utilisation.
usage: stream_bench.py SIZE CASE[,CASE...]   CASE in {valid, holes15, blobs15, sigma2, sigma3}
env: PBS=0,4  BATCHES=16,32  PAIRS=96  CSGPU_LIB"""
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
cases = (sys.argv[2] if len(sys.argv) > 2 else "valid,holes15").split(",")
P = int(os.environ.get("PAIRS", "96"))


def raster(case):
    rng = np.random.default_rng(11)
    z = rng.standard_normal((N, N))
    u = rng.random((N, N))
    if case == "valid":
        return np.exp(z)
    if case == "holes15":
        return np.where(u < 0.15, 0.0, np.exp(z))
    if case == "blobs15":
        k = max(4, N // 64)
        coarse = np.random.default_rng(3).standard_normal((N // k + 2, N // k + 2))
        field = np.kron(coarse, np.ones((k, k)))[:N, :N]
        return np.where(field < np.quantile(field, 0.15), 0.0, np.exp(z))
    if case.startswith("sigma"):
        return np.exp(float(case[5:]) * z)
    raise SystemExit("unknown case " + case)


for case in cases:
    g = raster(case)
    for pb in [int(v) for v in os.environ.get("PBS", "0,4").split(",")]:
        for B in [int(v) for v in os.environ.get("BATCHES", "16,32").split(",")]:
            with L.raster_setup(g, L.default_opts(batch=B, precond_bytes=pb)) as h:
                info = h.info
                labels, _ = h.components()
                pool = np.flatnonzero(labels == np.bincount(labels).argmax())
                pts = np.random.default_rng(5).choice(pool, size=15, replace=False)
                pairs = [(int(pts[i]), int(pts[j])) for i in range(15) for j in range(i + 1, 15)][:P]
                src, dst = [p[0] for p in pairs], [p[1] for p in pairs]
                h.solve_pairs(src[:B], dst[:B])          # warm-up (work vectors, code objects)
                ref = None
                modes = (("batch", {"CSGPU_NO_STREAM": "1"}), ("stream", {"CSGPU_STREAM": "1"}), ("adaptive", {}))
                if os.environ.get("REPEAT"):   # run-to-run reproducibility probe: every mode twice
                    modes = (modes[0], ("batch#2", modes[0][1]), modes[1], ("stream#2", modes[1][1]), ("batch#3", modes[0][1]))
                if os.environ.get("MODES"):
                    modes = tuple(m for m in modes if m[0] in os.environ["MODES"].split(","))
                for mode, env in modes:
                    for k in ("CSGPU_NO_STREAM", "CSGPU_STREAM"):
                        os.environ.pop(k, None)
                    os.environ.update(env)
                    t0 = time.perf_counter()
                    R, _, _, st = h.solve_pairs(src, dst)
                    ms = (time.perf_counter() - t0) * 1e3
                    if ref is None:
                        ref = R
                    slots = st["stream_slots"]
                    print(json.dumps({"case": case, "N": N, "precond_bytes": info["precond_bytes"], "batch": B, "mode": mode,
                                      "pairs": len(src), "ms_per_16_pairs": ms * 16.0 / len(src), "ms_total": ms,
                                      "iters_mean": st["total_iters"] / float(len(src)), "iters_max": st["max_iters"],
                                      "stream_slots": slots,
                                      "column_utilisation": (st["total_iters"] + len(src)) / float(slots * B) if slots else None,
                                      "not_converged": st["not_converged"], "max_relres": st["max_relres"],
                                      "identical_to_batch": bool(np.array_equal(R, ref)),
                                      "max_rel_diff_vs_batch": float(np.max(np.abs(R - ref) / np.abs(ref))),
                                      "lattice_period": info["lattice_period"], "levels": info["levels"]}), flush=True)
for k in ("CSGPU_NO_STREAM", "CSGPU_STREAM"):
    os.environ.pop(k, None)
