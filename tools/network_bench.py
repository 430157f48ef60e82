"""BASELINE config 5 (scaled): random network, advanced one-to-all semantics (scope row N2).
n nodes, 10*n undirected edges (endpoints from rng(424242), deduplicated, giant component kept), conductances U(0.5, 2);
K focal nodes; for each of S sources: unit current at the source, the other K-1 focal nodes tied to ground (rows
deleted, src/raster/advanced.jl:282-288) => S independent grounded solves with DIFFERENT matrices, each = one
csgpu_setup + one csgpu_solve_rhs (multiple_solve, advanced.jl:307-312).
Usage: python tools/network_bench.py [n] [sources] [--emu]"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.csgraph as csg

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import circuitscape_jl_amd  # noqa: F401,E402
from circuitscape_jl_amd import lib  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 100000
S = int(args[1]) if len(args) > 1 else 4
if "--emu" in sys.argv:
    lib.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "emu", "libcsgpu_emu.so"))
rng = np.random.default_rng(424242)
m = 10 * n
i = rng.integers(0, n, size=m)
j = rng.integers(0, n, size=m)
keep = i != j
lo, hi = np.minimum(i[keep], j[keep]), np.maximum(i[keep], j[keep])
key = np.unique(lo.astype(np.int64) * n + hi)
lo, hi = key // n, key % n
w = rng.uniform(0.5, 2.0, size=len(lo))
A = sp.coo_matrix((w, (lo, hi)), shape=(n, n)).tocsr()
A = (A + A.T).tocsr()
nc, lab = csg.connected_components(A, directed=False)
giant = np.flatnonzero(lab == np.bincount(lab).argmax())
A = A[giant][:, giant]
n = A.shape[0]
G = (sp.diags(np.asarray(A.sum(axis=1)).ravel()) - A).tocsr()
K = int(os.environ.get("NFOCAL", "8"))
focal = rng.choice(n, size=K, replace=False)
rows = []
if "--shared" in sys.argv:
    # round 2: ONE setup of the ungrounded Laplacian, every source a column of csgpu_solve_grounded (the other K-1 focal
    # nodes are that column's Dirichlet set), batches of 16 sources per PCG
    t0 = time.perf_counter()
    h = lib.setup(G, lib.default_opts(batch=16, precond_bytes=4, itmax=2000), index_dtype=np.int32, index_base=0)
    t1 = time.perf_counter()
    ns = min(S, K)
    B = np.zeros((n, ns))
    grounds = []
    for s in range(ns):
        B[focal[s], s] = 1.0
        grounds.append([int(q) for q in focal if q != focal[s]])
    X, _, st = h.solve_grounded(B, grounds)
    t2 = time.perf_counter()
    info = h.info
    worst = 0.0
    for s in range(ns):
        keepn = np.setdiff1d(np.arange(n), grounds[s])
        r = (G @ X[:, s] - B[:, s])[keepn]
        worst = max(worst, float(np.linalg.norm(r) / np.linalg.norm(B[keepn, s])))
    parity = None
    if "--check" in sys.argv:
        # independent check at scale: the reduced systems of the first two sources solved by scipy's CG (Jacobi
        # preconditioner, true-residual 1e-12) on the host; compared: the voltage at the source = its resistance to the
        # grounded focal nodes, and the whole voltage vector
        import scipy.sparse.linalg as spla
        errs, verrs = [], []
        for s in range(min(2, ns)):
            keepn = np.setdiff1d(np.arange(n), grounds[s])
            M = G[keepn][:, keepn].tocsr()
            b = B[keepn, s]
            dinv = 1.0 / M.diagonal()
            xs, flag = spla.cg(M, b, rtol=1e-12, atol=0.0, maxiter=2000, M=spla.LinearOperator(M.shape, lambda v: dinv * v))
            assert flag == 0 and np.linalg.norm(M @ xs - b) <= 1e-11 * np.linalg.norm(b)
            k = np.searchsorted(keepn, focal[s])
            errs.append(abs(X[focal[s], s] - xs[k]) / abs(xs[k]))
            verrs.append(float(np.max(np.abs(X[keepn, s] - xs)) / np.max(np.abs(xs))))
        parity = {"checker": "scipy CG + Jacobi, true residual 1e-12, reduced systems of the first two sources",
                  "max_rel_err_resistance": float(max(errs)), "max_rel_err_voltages": float(max(verrs))}
    print(json.dumps({"mode": "shared hierarchy (csgpu_solve_grounded)", "parity": parity,
                      "preconditioner": "AMG" if info["levels"] > 1 else "Jacobi (expander bail-out: no coarse level; amg_setup.h)",
                      "n": int(n), "nnz": int(G.nnz), "sources": ns,
                      "focal_nodes": K, "levels": info["levels"], "setup_wall_s": t1 - t0, "setup_device_ms": info["setup_ms"],
                      "upload_ms": info["upload_ms"], "solve_wall_s_all_sources": t2 - t1,
                      "per_source_s": (t2 - t0) / ns, "iters_mean": st["total_iters"] / ns, "worst_relres": worst}), flush=True)
    h.close()
    sys.exit(0)
for s in range(min(S, K)):
    src = focal[s]
    ground = np.setdiff1d(focal, [src])
    keepn = np.setdiff1d(np.arange(n), ground)
    M = G[keepn][:, keepn].tocsr()
    b = np.zeros(len(keepn))
    b[np.searchsorted(keepn, src)] = 1.0
    t0 = time.perf_counter()
    h = lib.setup(M, lib.default_opts(batch=1, precond_bytes=4, itmax=2000), index_dtype=np.int32, index_base=0)
    t1 = time.perf_counter()
    x, st = h.solve_rhs(b)
    t2 = time.perf_counter()
    info = h.info
    r = np.linalg.norm(M @ x - b) / np.linalg.norm(b)
    rows.append({"source": int(s), "n": int(M.shape[0]), "nnz": int(M.nnz), "levels": info["levels"],
                 "level_n": info["level_n"][:info["levels"]], "level_nnz": info["level_nnz"][:info["levels"]],
                 "operator_complexity": info["operator_complexity"], "setup_wall_s": t1 - t0,
                 "setup_device_ms": info["setup_ms"], "upload_ms": info["upload_ms"], "solve_wall_s": t2 - t1,
                 "iters": st["max_iters"], "relres": r, "polished": st["polished_batches"]})
    print(json.dumps(rows[-1]), flush=True)
    h.close()
