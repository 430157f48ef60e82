"""Numerical prototype (numpy/scipy) of the GPU AMG variant, used to choose parameters before writing kernels.
Not part of the product or the tests."""
import sys, time
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, "/root/repo")

def hash32(x):
    x = np.asarray(x, dtype=np.uint64)
    x = (x ^ (x >> 16)) * np.uint64(0x45d9f3b) & np.uint64(0xffffffff)
    x = (x ^ (x >> 16)) * np.uint64(0x45d9f3b) & np.uint64(0xffffffff)
    x = x ^ (x >> 16)
    return x.astype(np.uint64)

def mis2_aggregate(S):
    """S: symmetric strength pattern CSR without diagonal. Returns agg ids (n,), nagg."""
    n = S.shape[0]
    # state: 0 undecided, 1 in MIS, 2 removed ; key = (state_priority, hash, id)
    pri = (hash32(np.arange(n)) << np.uint64(32)) | np.arange(n, dtype=np.uint64)
    state = np.zeros(n, dtype=np.int8)
    indptr, indices = S.indptr, S.indices
    rows = np.repeat(np.arange(n), np.diff(indptr))
    it = 0
    while (state == 0).any():
        it += 1
        # undecided nodes carry their priority, MIS nodes carry +inf, removed carry 0
        key = np.where(state == 0, pri, np.where(state == 1, np.uint64(0xffffffffffffffff), np.uint64(0)))
        m1 = key.copy()
        np.maximum.at(m1, rows, key[indices])
        m2 = m1.copy()
        np.maximum.at(m2, rows, m1[indices])
        newmis = (state == 0) & (m2 == pri)
        state[newmis] = 1
        # remove undecided nodes within distance 2 of a MIS node
        mis = (state == 1).astype(np.int8)
        d1 = mis.copy(); np.maximum.at(d1, rows, mis[indices])
        d2 = d1.copy(); np.maximum.at(d2, rows, d1[indices])
        state[(state == 0) & (d2 == 1)] = 2
    roots = np.flatnonzero(state == 1)
    nagg = len(roots)
    agg = -np.ones(n, dtype=np.int64)
    agg[roots] = np.arange(nagg)
    # pass 1: neighbours of roots
    a1 = agg.copy()
    # each non-root picks the root neighbour (unique by MIS-2 property? dist-1 nbrs of two roots impossible) 
    cand = np.where(agg[indices] >= 0, agg[indices], -1)
    best = -np.ones(n, dtype=np.int64)
    np.maximum.at(best, rows, cand)
    a1 = np.where(agg >= 0, agg, best)
    # pass 2: remaining join strongest-connected aggregated neighbour
    rem = a1 < 0
    w = np.abs(S.data)
    # choose neighbour with max weight among aggregated
    wkey = np.where(a1[indices] >= 0, w, -1.0)
    bestw = -np.ones(n); np.maximum.at(bestw, rows, wkey)
    pick = (wkey == bestw[rows]) & (wkey >= 0)
    a2 = a1.copy()
    cand2 = np.where(pick, a1[indices], -1)
    b2 = -np.ones(n, dtype=np.int64); np.maximum.at(b2, rows, cand2)
    a2[rem] = b2[rem]
    return a2, nagg, it

def build_hierarchy(A, max_coarse=200, max_levels=12, omega=4/3, theta=0.0):
    levels = []
    B = np.ones(A.shape[0])
    while A.shape[0] > max_coarse and len(levels) + 1 < max_levels:
        n = A.shape[0]
        d = A.diagonal()
        S = A.copy().tocsr(); S.setdiag(0); S.eliminate_zeros()
        if theta > 0:
            rows = np.repeat(np.arange(n), np.diff(S.indptr))
            keep = S.data**2 >= theta**2 * np.abs(d[rows] * d[S.indices])
            S = sp.csr_matrix((S.data[keep], S.indices[keep], np.concatenate([[0], np.cumsum(np.bincount(rows[keep], minlength=n))])), shape=S.shape)
        agg, nagg, it = mis2_aggregate(S)
        assert (agg >= 0).all()
        nrm = np.sqrt(np.bincount(agg, weights=B * B, minlength=nagg))
        T = sp.csr_matrix((B / nrm[agg], (np.arange(n), agg)), shape=(n, nagg))
        L = np.asarray(abs(A).sum(axis=1)).ravel()
        P = (T - sp.diags(omega / L) @ (A @ T)).tocsr()
        R = P.T.tocsr()
        Ac = (R @ A @ P).tocsr()
        levels.append((A, P, R))
        A = Ac; B = nrm
    return levels, A

class VC:
    def __init__(self, levels, Ac, smoother="jacobi", nu=1, w=None, cheb_deg=2, cheb_ratio=4.0):
        self.levels = levels
        self.smoother = smoother; self.nu = nu; self.cheb_deg = cheb_deg
        Ad = Ac.toarray()
        ev, V = np.linalg.eigh(Ad)
        keep = ev > 1e-12 * ev.max()
        self.pinv = (V[:, keep] / ev[keep]) @ V[:, keep].T
        self.dinv = []; self.lmax = []; self.w = []
        for (A, P, R) in levels:
            d = A.diagonal()
            self.dinv.append(1.0 / d)
            L = np.asarray(abs(A).sum(axis=1)).ravel()
            rho = (L / d).max()   # gershgorin bound on rho(D^-1 A)
            self.lmax.append(rho)
            self.w.append((4.0 / 3.0) / rho if w is None else w)
        self.cheb_ratio = cheb_ratio
    def smooth(self, l, x, b, zero_guess):
        A = self.levels[l][0]; dinv = self.dinv[l]
        if self.smoother == "jacobi":
            for k in range(self.nu):
                if zero_guess and k == 0:
                    x = self.w[l] * dinv * b
                else:
                    x = x + self.w[l] * dinv * (b - A @ x)
            return x
        elif self.smoother == "l1":
            L = np.asarray(abs(A).sum(axis=1)).ravel()
            for k in range(self.nu):
                x = x + (b - A @ x) / L if not (zero_guess and k == 0) else b / L
            return x
        else:  # chebyshev on D^-1 A, eigen interval [lmax/ratio, lmax]
            lmax = self.lmax[l]; lmin = lmax / self.cheb_ratio
            theta = 0.5 * (lmax + lmin); delta = 0.5 * (lmax - lmin)
            sigma = theta / delta
            rho = 1.0 / sigma
            r = b - A @ x if not zero_guess else b.copy()
            dvec = dinv * r / theta
            x = x + dvec
            for k in range(1, self.cheb_deg):
                rho_new = 1.0 / (2 * sigma - rho)
                r = r - A @ dvec
                dvec = rho_new * rho * dvec + 2 * rho_new / delta * (dinv * r)
                x = x + dvec
                rho = rho_new
            return x
    def cycle(self, l, b):
        if l == len(self.levels):
            return self.pinv @ b
        A, P, R = self.levels[l]
        x = self.smooth(l, np.zeros_like(b), b, True)
        r = b - A @ x
        xc = self.cycle(l + 1, R @ r)
        x = x + P @ xc
        x = self.smooth(l, x, b, False)
        return x
    def __call__(self, r):
        return self.cycle(0, r)

def pcg(A, b, M, rtol=1e-6, atol=None, itmax=500, true_res=False):
    atol = np.sqrt(np.finfo(float).eps) if atol is None else atol
    x = np.zeros_like(b); r = b.copy(); z = M(r); p = z.copy()
    gamma = r @ z
    rn = np.sqrt(gamma) if not true_res else np.linalg.norm(r)
    eps = atol + rtol * rn
    it = 0
    while rn > eps and it < itmax:
        Ap = A @ p
        alpha = gamma / (p @ Ap)
        x += alpha * p; r -= alpha * Ap
        z = M(r)
        g2 = r @ z
        rn = np.sqrt(abs(g2)) if not true_res else np.linalg.norm(r)
        beta = g2 / gamma; gamma = g2
        p = z + beta * p
        it += 1
    return x, it

if __name__ == "__main__":
    from oracle import refsolve as rs, refgraph as rg
    N = int(sys.argv[1]); sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    G, g = rg.synthetic_raster_problem(N, N, sigma=sigma)
    A = rs.regularize(G)
    t = time.time(); levels, Ac = build_hierarchy(A, max_coarse=int(sys.argv[3]) if len(sys.argv) > 3 else 200); print("setup", time.time() - t)
    sizes = [(l[0].shape[0], l[0].nnz) for l in levels] + [(Ac.shape[0], Ac.nnz)]
    print(sizes, "opcx", sum(s[1] for s in sizes) / sizes[0][1], "P nnz/row", [l[1].nnz / l[1].shape[0] for l in levels])
    rng = np.random.default_rng(67890)
    cells = rng.choice(N * N, size=5, replace=False)
    b = np.zeros(N * N); b[cells[0]] = -1; b[cells[1]] = 1
    for name, kw in [("jacobi nu1", dict(smoother="jacobi", nu=1)), ("jacobi nu2", dict(smoother="jacobi", nu=2)),
                     ("l1 nu1", dict(smoother="l1", nu=1)),
                     ("cheb2 r4", dict(smoother="cheb", cheb_deg=2, cheb_ratio=4)), ("cheb3 r8", dict(smoother="cheb", cheb_deg=3, cheb_ratio=8)),
                     ("cheb2 r10", dict(smoother="cheb", cheb_deg=2, cheb_ratio=10)), ("cheb3 r30", dict(smoother="cheb", cheb_deg=3, cheb_ratio=30))]:
        M = VC(levels, Ac, **kw)
        x, it = pcg(A, b, M)
        x2, it2 = pcg(A, b, M, rtol=1e-10, atol=0, true_res=True)
        print(f"{name:12s} iters(ref tol)={it:3d} R={x[cells[1]]-x[cells[0]]:.10f}  iters(true 1e-10)={it2:3d} R={x2[cells[1]]-x2[cells[0]]:.12f}")
