#!/usr/bin/env python3
"""Fuzzer for the raster paths (round 3): random small rasters -- 6..39 cells a side, log-normal sigma 0.5..3.5, 0..35 % NODATA,
whole NODATA rows / columns, 4- / 8-neighbour, averaged resistances or conductances, fp64 / fp32 hierarchy -- through
csgpu_raster_setup (cell space, strength-aware tiles, index-free pipeline) AND through csgpu_setup with node coordinates
(the Julia host path), two pairs each, against a direct solve of the component's grounded system (scipy). Found the
single-level fp32 pseudo-inverse defect fixed in dense_sym_pinv. The graph is built by oracle/refgraph.py (test
infrastructure; input generation and checking only). Both paths carry the reference's regularisation shift eps * norm(nzval) (core.jl:161).
usage: fuzz_rasters.py SEED NCASES    (env CSGPU_LIB: library to load, default the emulator build)"""
import os, sys, json, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import circuitscape_jl_amd  # noqa: E402,F401
from circuitscape_jl_amd import lib as L, solver as ps  # noqa: E402
from oracle import refgraph as rg  # noqa: E402
L.load(os.environ.get("CSGPU_LIB", os.path.join(ROOT, "tests", "emu", "libcsgpu_emu.so")))
seed0 = int(sys.argv[1]); ncase = int(sys.argv[2])
def direct_R(A, s, d):
    # the reference's problem is the FULL regularised system (non-singular thanks to the shift, core.jl:161), not the system
    # grounded at s: on a maze of conductances over ten decades the shift's leak is visible (5e-5 of R)
    n = A.shape[0]
    b = np.zeros(n); b[d] = 1.0; b[s] = -1.0
    x = spla.spsolve(A.tocsc(), b)
    return x[d] - x[s]
bad = 0
for case in range(ncase):
    if case and case % 20 == 0: print("# seed", seed0, "cases done", case, "bad", bad, flush=True)
    rng = np.random.default_rng(seed0 * 1000 + case)
    lo_, hi_ = int(os.environ.get("FUZZ_MIN", "6")), int(os.environ.get("FUZZ_MAX", "40"))
    R = int(rng.integers(lo_, hi_)); C = int(rng.integers(lo_, hi_))
    sigma = float(rng.choice([0.5, 1.0, 2.5, 3.5]))
    frac = float(rng.choice([0.0, 0.05, 0.2, 0.35]))
    four = bool(rng.integers(0, 2)); avg = bool(rng.integers(0, 2)); pb = int(rng.choice([0, 4]))
    g = np.exp(sigma * rng.standard_normal((R, C)))
    g[rng.random((R, C)) < frac] = 0.0
    if rng.random() < 0.3: g[rng.integers(0, R), :] = 0.0          # an all-NODATA row
    if rng.random() < 0.3: g[:, rng.integers(0, C)] = 0.0          # an all-NODATA column
    if (g > 0).sum() < 12: continue
    nm = rg.construct_node_map(g, None)
    W = rg.construct_graph(g, nm, avg, four)
    A = sp.csr_matrix(rg.laplacian(W))
    ncomp, lab = sp.csgraph.connected_components(W, directed=False)
    big = np.flatnonzero(lab == np.bincount(lab).argmax())
    if len(big) < 4: continue
    ids = rng.choice(big, size=4, replace=False)
    src, dst = [int(ids[0]), int(ids[1])], [int(ids[2]), int(ids[3])]
    loc0 = {int(v): k for k, v in enumerate(big)}
    Ab = sp.csr_matrix(A[big][:, big], copy=True)
    Ab.data = Ab.data + np.finfo(np.float64).eps * np.linalg.norm(Ab.data)   # the reference's shift (core.jl:161): part of the problem
    Rd = np.array([direct_R(Ab, loc0[s], loc0[d]) for s, d in zip(src, dst)])
    # csgpu_raster_setup shifts by eps * norm over the WHOLE raster's nonzeros (include/csgpu.h: one handle serves all
    # components), the reference -- and the host-CSR path below -- by the component's own norm (core.jl:158-161). On a sigma = 3.5
    # raster with two components the two norms differed 6x and a point-grounded solve moved 6e-6 (fuzz case 9111/80, round 5):
    # every check of the RASTER handle is made against the system with the raster-wide shift.
    Ar = sp.csr_matrix(A[big][:, big], copy=True)
    Ar.data = Ar.data + np.finfo(np.float64).eps * np.linalg.norm(A.data)
    Rd_r = np.array([direct_R(Ar, loc0[s], loc0[d]) for s, d in zip(src, dst)])
    tag = dict(case=case, R=R, C=C, sigma=sigma, frac=frac, four=four, avg=avg, pb=pb)
    try:
        with L.raster_setup(g, L.default_opts(batch=2, precond_bytes=pb, rtol=1e-10, atol=0.0), four_neighbors=four, avg_resistances=avg) as h:
            Rr, _, volt, st = h.solve_pairs(src, dst, want_voltages=True)
            e1 = float(np.max(np.abs(Rr - Rd_r) / Rd_r))
            lat = h.info["lattice_period"]
            # node currents, cumulative and maximum maps (N1) and a grounded solve (N2) on the same handle, against the
            # direct solve of the component
            from oracle import refmaps
            n_all = A.shape[0]
            cum = np.zeros(n_all); mx = np.zeros(n_all)
            Rc, _, cur, _ = h.solve_pairs_currents(src, dst, cum=cum, mx=mx)
            ec = 0.0
            exp_cum = np.zeros(n_all); exp_mx = np.zeros(n_all)
            for p_, (s_, d_) in enumerate(zip(src, dst)):
                b_ = np.zeros(len(big)); b_[loc0[d_]] = 1.0; b_[loc0[s_]] = -1.0
                v_ = spla.spsolve(Ar.tocsc(), b_)
                v_ = v_ - v_[loc0[s_]]
                nc_ = np.zeros(n_all); nc_[big] = refmaps.get_node_currents(Ar, v_)
                ec = max(ec, float(np.max(np.abs(cur[:, p_] - nc_)) / max(nc_.max(), 1e-300)))
                exp_cum += nc_; exp_mx = np.maximum(exp_mx, nc_)
            ec = max(ec, float(np.max(np.abs(cum - exp_cum)) / exp_cum.max()), float(np.max(np.abs(mx - exp_mx)) / exp_mx.max()))
            Bg = np.zeros((n_all, 1)); Bg[src[0], 0] = 1.0
            Xg, _, stg = h.solve_grounded(Bg, [[dst[0]]])
            keep_g = np.setdiff1d(np.arange(len(big)), [loc0[dst[0]]])   # Dirichlet at dst[0]: the grounded system
            bg_ = np.zeros(len(big)); bg_[loc0[src[0]]] = 1.0
            Rg = spla.spsolve(Ar[keep_g][:, keep_g].tocsc(), bg_[keep_g])[np.searchsorted(keep_g, loc0[src[0]])]
            eg = abs(Xg[src[0], 0] - Rg) / Rg
            if ec > 1e-5 or eg > 1e-6 or stg["not_converged"]:
                bad += 1
                print("BAD-MAPS", tag, ec, eg, flush=True)
        # host CSR path with coordinates (largest component)
        comp = big + 1
        Ac = A[big][:, big]
        Ac = sp.csr_matrix(Ac, copy=True)
        Ac.data = Ac.data + np.finfo(np.float64).eps * np.linalg.norm(Ac.data)   # the reference's shift (core.jl:161)
        row, col = ps._node_coords(nm, comp)
        loc = {int(v): k for k, v in enumerate(big)}
        with L.setup(sp.csr_matrix(Ac), L.default_opts(batch=2, precond_bytes=pb, rtol=1e-10, atol=0.0), node_row=row, node_col=col) as h2:
            R2, _, _, st2 = h2.solve_pairs([loc[s] for s in src], [loc[d] for d in dst])
            e2 = float(np.max(np.abs(R2 - Rd) / Rd))
            lat2 = h2.info["lattice_period"]
        ok = e1 < 1e-6 and e2 < 1e-6 and st["not_converged"] == 0 and st2["not_converged"] == 0
        if not ok:
            bad += 1
            print("BAD", tag, e1, e2, lat, lat2, st["total_iters"], st2["total_iters"], flush=True)
    except Exception as ex:
        bad += 1
        print("EXC", tag, str(ex)[:200], flush=True)
print("seed", seed0, "cases", ncase, "bad", bad)
