#!/usr/bin/env python3
"""BASELINE configs[3]'s precision AT ITS SIZE: the all-fp32 handle of the 10000 x 10000 bench raster
(`precision = single`, src/run.jl:29: Float32 Laplacian, every stored entry shifted by eps(Float32) * norm(nzval),
src/core.jl:161 -- the shift grows with n) with the library's (= the reference's) defaults, 16 pairs, against the TIGHT CPU
oracle run in double on the very fp32 matrix the device holds (downloaded from the handle). Writes the oracle's resistances
as a golden fixture (tests/golden/full_size_10000_fp32.json) and a log (profiles/r6_parity16_10000_fp32.json)."""
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
    ap.add_argument("--pairs", type=int, default=16)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r6_parity16_10000_fp32.json"))
    ap.add_argument("--fixture", default=os.path.join(ROOT, "tests", "golden", "full_size_10000_fp32.json"))
    ap.add_argument("--oracle-rtol", type=float, default=1e-10)
    args = ap.parse_args()
    import bench
    import circuitscape_jl_amd  # noqa: F401
    from circuitscape_jl_amd import lib
    from oracle import refsolve as rs
    lib.load(os.environ.get("CSGPU_LIB"))
    N, npairs = args.size, args.pairs
    g32 = bench.make_raster(N, dtype=np.float32)
    _, pairs = bench.focal_pairs(N)
    src = [p[0] for p in pairs[:npairs]]
    dst = [p[1] for p in pairs[:npairs]]
    h = lib.raster_setup(g32, lib.default_opts(batch=16))          # all-fp32 handle, reference defaults
    info = h.info
    R, _, _, st = h.solve_pairs(src, dst)
    t0 = time.time()
    A = h.level_matrix(0, "A")
    t_dl = time.time() - t0
    h.close()
    assert A.dtype == np.float32
    diag = A.diagonal().astype(np.float64)
    A = A.astype(np.float64)                                        # (exact: every fp32 value is an fp64 value)
    out = {"size": N, "n": int(A.shape[0]), "nnz": int(A.nnz), "pairs": npairs, "download_s": t_dl,
           "gpu": {"R": [float(x) for x in R], "iters_mean": st["total_iters"] / float(npairs), "max_relres": st["max_relres"],
                   "not_converged": st["not_converged"], "levels": info["levels"], "lattice_period": info["lattice_period"]},
           # the regularisation as the device applied it: every off-diagonal entry is -w + shift, the row sums are 9 x shift
           # (interior rows); reported so that a reader can compare with eps(Float32) * norm(nzval)
           "row_sum_interior": float(np.asarray(A[N + 1].sum())), "eps32": float(np.finfo(np.float32).eps),
           "diag_min_max": [float(diag.min()), float(diag.max())]}
    json.dump(out, open(args.out, "w"), indent=1)
    t0 = time.time()
    S = rs.OracleAMG(A)
    out["oracle"] = {"setup_s": time.time() - t0, "levels": S.levels}
    t0 = time.time()
    Ro, _, r = S.solve_pairs(src, dst, rtol=args.oracle_rtol, atol=0.0, criterion=1, nthreads=npairs)
    out["oracle"].update({"solve_s": time.time() - t0, "R": [float(x) for x in Ro], "iters": [x["iters"] for x in r],
                          "true_relres": [x["true_relres"] for x in r], "rtol_true_residual": args.oracle_rtol})
    rel = float(np.max(np.abs(R - Ro) / np.abs(Ro)))
    out.update({"max_rel_err_vs_oracle": rel, "tolerance": 1e-4, "ok": bool(rel < 1e-4)})
    fx = {"what": "BASELINE.json configs[3] precision at the BASELINE size: bench.make_raster(%d, dtype=float32) (r = exp(N(0,1)), "
                  "seed 12345, g = 1/r; 8-neighbour, average conductance), csgpu_raster_setup with val_bytes = 4 and the "
                  "library's (= the reference's) defaults -- every stored entry shifted by eps(Float32) * norm(nzval), "
                  "src/core.jl:161 -- effective resistances of the first %d pairs of bench.focal_pairs(%d) from the TIGHT CPU "
                  "oracle (oracle/cs_oracle.cpp, true-residual rtol %g) run in double on the very fp32 matrix the GPU handle "
                  "holds (downloaded from the handle)" % (N, npairs, N, args.oracle_rtol),
          "generated_by": "tools/full_size_fp32.py on the GPU box's host cores (oracle set-up %.0f s + %.0f s for the %d pairs "
                          "on %d threads)" % (out["oracle"]["setup_s"], out["oracle"]["solve_s"], npairs, npairs),
          "size": N, "pairs": [[int(a), int(b)] for a, b in zip(src, dst)], "R_tight": [float(x) for x in Ro],
          "oracle_true_relres": [float(x["true_relres"]) for x in r], "oracle_iters": [int(x["iters"]) for x in r],
          "row_sum_interior": out["row_sum_interior"], "tolerance_rel": 1e-4}
    json.dump(fx, open(args.fixture, "w"), indent=1)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k not in ("oracle", "gpu")}))


if __name__ == "__main__":
    main()
