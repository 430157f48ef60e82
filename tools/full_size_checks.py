#!/usr/bin/env python3
"""One-off checks at the BASELINE size (10000 x 10000, n = 1e8) that are too heavy for the test-suite or the bench:

 1. parity: the GPU path with bench.py's defaults (batch 16, fp32 preconditioner, lattice-form CG product) and the
    all-fp64 path against the TIGHT CPU oracle (true-residual stopping rule) on the first `--pairs` pairs of the bench's
    pair list. The oracle is handed the very matrix the GPU solves with (downloaded from the handle).
 2. the integration path a Julia host uses: csgpu_setup from Int64 / 1-based host CSR arrays (upload + index conversion
    + lattice detection + AMG setup), timed, and one batch solved through that handle.

Writes one JSON object (default profiles/r2_parity_10000.json). Needs ~60 GB of host memory at 10000^2.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=10000)
    ap.add_argument("--pairs", type=int, default=2)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r2_parity_10000.json"))
    ap.add_argument("--oracle-rtol", type=float, default=1e-10)
    ap.add_argument("--skip-oracle", type=int, default=0)
    ap.add_argument("--skip-host-csr", type=int, default=0)
    ap.add_argument("--fixture", default="",
                    help="also write the tight oracle's resistances as a golden fixture (tests/golden/full_size_10000.json "
                         "layout) to this path")
    args = ap.parse_args()
    import bench
    import circuitscape_jl_amd  # noqa: F401
    from circuitscape_jl_amd import lib
    lib.load(os.environ.get("CSGPU_LIB"))
    N = args.size
    out = {"size": N, "n": N * N}
    mem_gb = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                mem_gb = int(line.split()[1]) / 1e6
    except Exception:
        pass
    out["host_mem_available_gb"] = mem_gb
    g = bench.make_raster(N)
    cells, pairs = bench.focal_pairs(N)
    src = [p[0] for p in pairs[:16]]
    dst = [p[1] for p in pairs[:16]]
    res = {}
    A = None
    for name, pb in (("mixed", 4), ("fp64", 0)):
        h = lib.raster_setup(g, lib.default_opts(batch=16, precond_bytes=pb))
        R, _, _, st = h.solve_pairs(src, dst)
        res[name] = R
        out["gpu_" + name] = {"R": R[:args.pairs].tolist(), "iters_mean": st["total_iters"] / 16.0,
                              "max_relres": st["max_relres"], "lattice_period": h.info["lattice_period"]}
        if name == "fp64":
            t0 = time.time()
            A = h.level_matrix(0, "A")
            out["download_s"] = time.time() - t0
        h.close()
    out["max_rel_diff_mixed_vs_fp64_16pairs"] = float(np.max(np.abs(res["mixed"] - res["fp64"]) / res["fp64"]))
    del g
    # ---- 2. host CSR entry point (Julia's arrays: Int64, 1-based)
    if not args.skip_host_csr:
      rp = np.ascontiguousarray(A.indptr.astype(np.int64) + 1)
      ci = np.ascontiguousarray(A.indices.astype(np.int64) + 1)
      va = np.ascontiguousarray(A.data, dtype=np.float64)
      t0 = time.perf_counter()
      h2 = lib.setup_arrays(rp, ci, va, A.shape[0], A.nnz, lib.default_opts(batch=16, precond_bytes=4), index_base=1)
      wall = time.perf_counter() - t0
      i2 = h2.info
      R2, _, _, st2 = h2.solve_pairs(src, dst)
      h2.close()
      host_bytes = int(rp.nbytes + ci.nbytes + va.nbytes)   # (before the arrays go: ADVICE r4)
      del rp, ci, va
      out["host_csr"] = {"setup_wall_s": wall, "upload_convert_s": i2["upload_ms"] / 1e3, "device_setup_s": i2["setup_ms"] / 1e3,
                       "host_bytes": host_bytes, "lattice_period_detected": i2["lattice_period"],
                       "levels": i2["levels"], "iters_mean": st2["total_iters"] / 16.0,
                       "max_rel_diff_R_vs_raster_entry_point": float(np.max(np.abs(R2 - res["mixed"]) / res["mixed"])),
                       "note": "no raster coordinates are handed over on this path: the lattice period detected from the "
                               "matrix supplies them (same 3x3-tile aggregation as the raster entry point)"}
    json.dump(out, open(args.out, "w"), indent=1)
    # ---- 1. oracle on the same matrix
    if not args.skip_oracle:
        from oracle import refsolve as rs
        t0 = time.time()
        S = rs.OracleAMG(A)
        out["oracle"] = {"setup_s": time.time() - t0, "levels": S.levels, "operator_complexity": S.operator_complexity}
        del A
        json.dump(out, open(args.out, "w"), indent=1)
        t0 = time.time()
        np_ = args.pairs
        Ro, _, r = S.solve_pairs(src[:np_], dst[:np_], rtol=args.oracle_rtol, atol=0.0, criterion=1, nthreads=np_)
        out["oracle"].update({"solve_s": time.time() - t0, "R": Ro.tolist(), "iters": [x["iters"] for x in r],
                              "true_relres": [x["true_relres"] for x in r], "rtol_true_residual": args.oracle_rtol})
        for name in ("mixed", "fp64"):
            out["max_rel_err_vs_oracle_" + name] = float(np.max(np.abs(res[name][:np_] - Ro) / Ro))
        out["tolerance"] = 1e-6
        out["ok"] = bool(max(out["max_rel_err_vs_oracle_mixed"], out["max_rel_err_vs_oracle_fp64"]) < 1e-6)
        if args.fixture:
            fx = {"what": "BASELINE.json configs[2] raster (bench.make_raster(%d): r = exp(N(0,1)), seed 12345, g = 1/r; "
                          "8-neighbour, average conductance, regularised like core.jl:161) -- effective resistances of the "
                          "first %d pairs of bench.focal_pairs(%d) (one full batch of 16) from the TIGHT CPU oracle "
                          "(oracle/cs_oracle.cpp, true-residual rtol %g) run on the very matrix the GPU handle holds "
                          "(downloaded from the handle)" % (N, np_, N, args.oracle_rtol),
                  "generated_by": "tools/full_size_checks.py --pairs %d --fixture ... on the GPU box's host cores "
                                  "(oracle setup %.0f s + %.0f s for the %d pairs on %d threads)"
                                  % (np_, out["oracle"]["setup_s"], out["oracle"]["solve_s"], np_, np_),
                  "size": N, "pairs": [[int(a), int(b)] for a, b in zip(src[:np_], dst[:np_])],
                  "R_tight": [float(x) for x in Ro], "oracle_true_relres": [float(x["true_relres"]) for x in r],
                  "oracle_iters": [int(x["iters"]) for x in r], "tolerance_rel": 1e-6}
            json.dump(fx, open(args.fixture, "w"), indent=1)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
