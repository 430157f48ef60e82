#!/usr/bin/env python3
"""Fuzzer for the CSR paths (round 3): random graphs WITHOUT lattice structure -- Erdos-Renyi, preferential attachment,
random trees, paths, stars, rings with chords, barbells, several components -- with conductances over 0 .. 6 decades,
through csgpu_setup (the Julia host path of network problems, network/pairwise.jl:4-29): pair solves, a general right-hand
side on a grounded copy (multiple_solve semantics, raster/advanced.jl:282-312), grounded solves on the shared hierarchy and
node currents, fp64 / fp32 hierarchy, batch 1..5, against direct solves (scipy). The matrix carries the reference's
regularisation shift eps * norm(nzval) (core.jl:161). Test infrastructure only.
usage: fuzz_networks.py SEED NCASES    (env CSGPU_LIB: library to load, default the emulator build)"""
import os, sys, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import circuitscape_jl_amd  # noqa: E402,F401
from circuitscape_jl_amd import lib as L  # noqa: E402
from oracle import refmaps  # noqa: E402
L.load(os.environ.get("CSGPU_LIB", os.path.join(ROOT, "tests", "emu", "libcsgpu_emu.so")))
seed0 = int(sys.argv[1]); ncase = int(sys.argv[2])
NMIN, NMAX = int(os.environ.get("FUZZ_MIN", "5")), int(os.environ.get("FUZZ_MAX", "700"))


def gen_edges(rng, kind, n):
    if kind == "er":
        m = int(n * rng.uniform(1.0, 6.0))
        return rng.integers(0, n, m), rng.integers(0, n, m)
    if kind == "pa":      # preferential attachment: hubs (long rows)
        k = int(rng.integers(1, 4)); targ = [0]; I = []; J = []
        for v in range(1, n):
            for t in rng.choice(targ, size=min(k, len(targ)), replace=False):
                I.append(v); J.append(int(t)); targ += [int(t), v]
        return np.array(I), np.array(J)
    if kind == "tree":
        return np.arange(1, n), np.array([rng.integers(0, v) for v in range(1, n)])
    if kind == "path":
        return np.arange(n - 1), np.arange(1, n)
    if kind == "star":
        return np.zeros(n - 1, dtype=np.int64), np.arange(1, n)
    if kind == "ring":
        m = int(rng.integers(0, max(1, n // 4)))
        return (np.concatenate([np.arange(n), rng.integers(0, n, m)]),
                np.concatenate([(np.arange(n) + 1) % n, rng.integers(0, n, m)]))
    if kind == "barbell":  # two cliques joined by a path: a weak link between dense blocks
        c = max(3, n // 3); I = []; J = []
        for a in range(c):
            for b in range(a + 1, c):
                I += [a, n - 1 - a]; J += [b, n - 1 - b]
        for v in range(c - 1, n - c):
            I.append(v); J.append(v + 1)
        return np.array(I), np.array(J)
    raise ValueError(kind)


def direct(A, b, ground):
    n = A.shape[0]
    keep = np.setdiff1d(np.arange(n), ground)
    x = np.zeros(n)
    x[keep] = spla.spsolve(A[keep][:, keep].tocsc(), b[keep])
    return x


bad = 0
for case in range(ncase):
    if case and case % 20 == 0: print("# seed", seed0, "cases done", case, "bad", bad, flush=True)
    rng = np.random.default_rng(seed0 * 1000 + case)
    kind = str(rng.choice(["er", "pa", "tree", "path", "star", "ring", "barbell"]))
    n = int(rng.integers(NMIN, NMAX))
    decades = float(rng.choice([0.0, 1.0, 3.0, 6.0]))
    pb = int(rng.choice([0, 4])); batch = int(rng.integers(1, 6))
    I, J = gen_edges(rng, kind, n)
    ok_ = I != J
    I, J = I[ok_], J[ok_]
    if len(I) == 0: continue
    w = 10.0 ** (decades * (rng.random(len(I)) - 0.5))
    if rng.random() < 0.3:  # cut the graph into components
        cut = rng.random(len(I)) < 0.25
        I, J, w = I[~cut], J[~cut], w[~cut]
        if len(I) == 0: continue
    W = sp.coo_matrix((w, (I, J)), shape=(n, n)).tocsr()
    W = W + W.T
    ncomp, lab = sp.csgraph.connected_components(W, directed=False)
    big = np.flatnonzero(lab == np.bincount(lab).argmax())
    if len(big) < 4: continue
    Wb = sp.csr_matrix(W[big][:, big])
    A = sp.csr_matrix(sp.diags(np.asarray(Wb.sum(axis=1)).ravel()) - Wb)
    A.sort_indices()
    A.data = A.data + np.finfo(np.float64).eps * np.linalg.norm(A.data)
    nb = len(big)
    npair = int(rng.integers(1, 8))
    src = [int(v) for v in rng.integers(0, nb, npair)]
    dst = [int(v) for v in rng.integers(0, nb, npair)]
    tag = dict(case=case, kind=kind, n=nb, nnz=int(A.nnz), decades=decades, pb=pb, batch=batch, npair=npair)
    try:
        Rd = np.zeros(npair); Vd = []
        for p, (s, d) in enumerate(zip(src, dst)):
            b = np.zeros(nb); b[d] += 1.0; b[s] -= 1.0
            # the reference's problem is the FULL regularised system (non-singular thanks to the shift, core.jl:161),
            # voltages shifted to v[src] = 0 afterwards (core.jl:231)
            v = spla.spsolve(A.tocsc(), b) if s != d else np.zeros(nb)
            v = v - v[s]
            Rd[p] = v[d]; Vd.append(v)
        with L.setup(A, L.default_opts(batch=batch, precond_bytes=pb, rtol=1e-10, atol=0.0)) as h:
            R, _, volt, st = h.solve_pairs(src, dst, want_voltages=True)
            scale = max(np.max(np.abs(Rd)), 1e-300)
            e1 = float(np.max(np.abs(R - Rd)) / scale)
            ev = max(float(np.max(np.abs(volt[:, p] - Vd[p])) / scale) for p in range(npair))
            # true relative residual of the returned voltages: separates conditioning (tiny residual, error = cond * residual)
            # from defects
            r1 = 0.0
            for p, (s, d) in enumerate(zip(src, dst)):
                if s == d: continue
                b = np.zeros(nb); b[d] += 1.0; b[s] -= 1.0
                r_ = A @ volt[:, p] - b
                r_ -= (A @ np.ones(nb)) * (r_.sum() / (A @ np.ones(nb)).sum())   # modulo the grounding shift
                r1 = max(r1, float(np.linalg.norm(r_) / np.linalg.norm(b)))
            # node currents of the same pairs (N1)
            Rc, _, cur, _ = h.solve_pairs_currents(src, dst)
            ec = 0.0
            for p in range(npair):
                if src[p] == dst[p]: continue
                nc = refmaps.get_node_currents(A, Vd[p])
                ec = max(ec, float(np.max(np.abs(cur[:, p] - nc)) / max(nc.max(), 1e-300)))
            # grounded solves on the shared hierarchy (N2): unit source, a random ground set per column
            ncol = int(rng.integers(1, 4))
            B = np.zeros((nb, ncol)); grounds = []; Xd = np.zeros((nb, ncol))
            for c in range(ncol):
                gs = np.unique(rng.integers(0, nb, int(rng.integers(1, 4))))
                s_ = int(rng.integers(0, nb))
                while s_ in gs: s_ = int(rng.integers(0, nb))
                B[s_, c] = 1.0; grounds.append([int(v) for v in gs])
                Xd[:, c] = direct(A, B[:, c], gs)
            Xg, _, stg = h.solve_grounded(B, grounds)
            eg = float(np.max(np.abs(Xg - Xd)) / max(np.max(np.abs(Xd)), 1e-300))
            rg_ = 0.0
            for c in range(ncol):
                keep = np.setdiff1d(np.arange(nb), grounds[c])
                rg_ = max(rg_, float(np.linalg.norm((A @ Xg[:, c] - B[:, c])[keep]) / np.linalg.norm(B[keep, c])))
            levels = h.info["levels"]
        # multiple_solve semantics: a grounded (SPD) matrix with a general right-hand side, its own hierarchy
        gd = rng.random(nb) < 0.1
        gd[int(rng.integers(0, nb))] = True
        Ag = sp.csr_matrix(A + sp.diags(gd * 10.0 ** rng.uniform(-2, 2)))
        bg = rng.standard_normal(nb)
        xd = spla.spsolve(Ag.tocsc(), bg)
        with L.setup(Ag, L.default_opts(batch=batch, precond_bytes=pb, rtol=1e-10, atol=0.0)) as h3:
            xg, st3 = h3.solve_rhs(bg)
            e3 = float(np.max(np.abs(xg - xd)) / max(np.max(np.abs(xd)), 1e-300))
            r3 = float(np.linalg.norm(Ag @ xg - bg) / np.linalg.norm(bg))
        # a solution whose true residual is at the level the tolerance asks for is right up to the conditioning of the
        # matrix (conductances over six decades: error = cond * residual can exceed 1e-6); a defect shows in both
        ok = ((e1 < 1e-6 and ev < 1e-6 or r1 < 1e-9) and (ec < 1e-5 or r1 < 1e-9) and (eg < 1e-6 or rg_ < 1e-9)
              and (e3 < 1e-6 or r3 < 1e-9) and st["not_converged"] == 0
              and stg["not_converged"] == 0 and st3["not_converged"] == 0)
        if not ok:
            bad += 1
            print("BAD", tag, dict(e1=e1, ev=ev, r1=r1, ec=ec, eg=eg, rg=rg_, e3=e3, r3=r3, levels=levels, it=st["total_iters"], itg=stg["total_iters"]), flush=True)
    except Exception as ex:
        bad += 1
        print("EXC", tag, str(ex)[:300], flush=True)
print("seed", seed0, "cases", ncase, "bad", bad)
