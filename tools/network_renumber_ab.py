#!/usr/bin/env python3
"""Does a locality-restoring renumbering of a NETWORK at set-up pay? A random geometric graph (mean
degree 10) whose node ids carry no locality -- what network mode gets from a file (src/network/pairwise.jl:31-65) --
solved (a) as given, (b) renumbered on the host by scipy's reverse Cuthill-McKee, (c) renumbered along the true
coordinates (cells of a sqrt(n)/8 grid, row-major: the best a renumbering could do). Per variant: set-up, 16 pair solves
from one anchor at batch 16, the CSR SpMM's time / GB/s (algorithmic bytes), iterations. One JSON line per variant."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import circuitscape_jl_amd  # noqa: F401,E402
from circuitscape_jl_amd import lib  # noqa: E402


def main():
    import scipy.sparse.csgraph as csg
    from scipy.spatial import cKDTree
    import torch
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    lib.load(os.environ.get("CSGPU_LIB"))
    dev = torch.device("cuda", 0) if torch.cuda.is_available() else None
    rng = np.random.default_rng(777)
    pts = rng.random((n, 2))
    pr = cKDTree(pts).query_pairs(np.sqrt(10.0 / (np.pi * n)), output_type="ndarray")
    lo, hi = np.minimum(pr[:, 0], pr[:, 1]).astype(np.int64), np.maximum(pr[:, 0], pr[:, 1]).astype(np.int64)
    w = rng.uniform(0.5, 2.0, size=len(lo))
    G = bench._csr_laplacian_from_edges(lo, hi, w, n, torch if dev is not None else None, dev)
    _, lab = csg.connected_components(G, directed=False)
    giant = np.flatnonzero(lab == np.bincount(lab).argmax())
    G = G[giant][:, giant].tocsr()
    G.sort_indices()
    xy = pts[giant]
    m = G.shape[0]
    eps = np.finfo(np.float64).eps * np.sqrt(np.dot(G.data, G.data))
    G.data += eps                                   # (core.jl:161)
    focal = rng.choice(m, size=17, replace=False)
    variants = {"as given (ids without locality)": np.arange(m)}
    t0 = time.time()
    variants["reverse Cuthill-McKee (scipy, host)"] = np.asarray(csg.reverse_cuthill_mckee(G, symmetric_mode=True))
    t_rcm = time.time() - t0
    side = max(1, int(np.sqrt(m) / 8))
    cell = (np.minimum((xy[:, 1] * side).astype(np.int64), side - 1) * side + np.minimum((xy[:, 0] * side).astype(np.int64), side - 1))
    variants["sorted by the true coordinates (grid cells, row-major)"] = np.argsort(cell, kind="stable")
    ref = None
    for name, perm in variants.items():
        inv = np.empty(m, dtype=np.int64)
        inv[perm] = np.arange(m)
        A = G[perm][:, perm].tocsr()
        A.sort_indices()
        src = [int(inv[focal[0]])] * 16
        dst = [int(inv[q]) for q in focal[1:]]
        t0 = time.perf_counter()
        h = lib.setup(A, lib.default_opts(batch=16, precond_bytes=0), index_dtype=np.int32, index_base=0)
        t_setup = time.perf_counter() - t0
        info = h.info
        h.solve_pairs(src, dst)
        t0 = time.perf_counter()
        R, _, _, st = h.solve_pairs(src, dst)
        wall = time.perf_counter() - t0
        k1 = h.spmv_bench(16, 10)
        h.close()
        if ref is None:
            ref = R
        calls = max(st["cg_spmv_calls"], 1)
        bw = np.mean(np.abs(A.indices - np.repeat(np.arange(m), np.diff(A.indptr))))
        print(json.dumps({"variant": name, "n": int(m), "nnz": int(A.nnz), "mean_abs_col_minus_row": float(bw),
                          "host_rcm_s": t_rcm if "Cuthill" in name else None, "levels": info["levels"],
                          "level_n": info["level_n"], "operator_complexity": info["operator_complexity"],
                          "setup_wall_s": t_setup, "setup_device_s": info["setup_ms"] / 1e3, "iters_mean": st["total_iters"] / 16.0,
                          "solve_wall_s": wall, "solve_device_s": st["device_ms"] / 1e3, "cg_spmm_ms": st["cg_spmv_ms"] / calls,
                          "cg_spmm_GBs": st["cg_spmv_bytes"] / (st["cg_spmv_ms"] / calls * 1e-3) / 1e9,
                          "spmm_k16_bench_ms": k1, "max_rel_diff_R_vs_first": float(np.max(np.abs(R - ref) / ref))}), flush=True)


if __name__ == "__main__":
    main()
