#!/usr/bin/env python3
"""scipy model of the library's hierarchy on a raster with NODATA cells (analysis tool of round 5; see README.md).
usage: model.py PICKLE {iters|spectrum|patterns|aggregation|enrich}"""
import pickle
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.csgraph as csg
import scipy.sparse.linalg as spla

OMEGA_S = 1.7


class Model:
    def __init__(self, path):
        d = pickle.load(open(path, "rb"))
        self.d = d
        mats, g, info = d["mats"], d["g"], d["info"]
        self.R, self.C = g.shape
        self.valid = (g.T.ravel() > 0)                          # column-major cell order
        self.n2c = np.flatnonzero(self.valid)
        self.ncell = self.valid.size
        nn = self.n2c.size
        E = sp.csr_matrix((np.ones(nn), (self.n2c, np.arange(nn))), shape=(self.ncell, nn))
        A0 = mats[(0, "A")].astype(np.float64)
        self.A0 = (E @ A0 @ E.T + sp.diags((~self.valid).astype(float))).tocsr()   # cell space: NODATA rows = identity
        self.L = info["levels"]
        self.A = [self.A0] + [mats[(l, "A")].astype(np.float64) for l in range(1, self.L)]
        self.P = [mats[(l, "P")].astype(np.float64) for l in range(self.L - 1)]
        self.w = []
        for l in range(self.L):
            rho = self.rho_gershgorin(self.A[l])
            self.w.append([OMEGA_S / rho] if l == 0 else list(self.chebyshev(rho, 2 if l == 1 else 3)))
        self.coarse_inv = np.linalg.pinv(self.A[-1].toarray())
        self.exact = {}
        R_, C_ = self.R, self.C
        self.Rc = R_ // 3 if R_ % 3 == 1 else (R_ + 2) // 3
        self.Cc = C_ // 3 if C_ % 3 == 1 else (C_ + 2) // 3
        rows, cols = np.arange(self.ncell) % R_, np.arange(self.ncell) // R_
        self.rows, self.cols = rows, cols
        self.tile = np.minimum(cols // 3, self.Cc - 1) * self.Rc + np.minimum(rows // 3, self.Rc - 1)
        labels = d["labels"]
        big = np.flatnonzero(labels == np.bincount(labels).argmax())
        self.giant_cells = self.n2c[big]
        self.ids = np.random.default_rng(5).choice(big, size=16, replace=False)

    @staticmethod
    def rho_gershgorin(M):
        return np.max(np.asarray(abs(M).sum(axis=1)).ravel() / M.diagonal())

    @staticmethod
    def chebyshev(rho, m, lo=0.1):
        a, b = rho * lo, rho
        k = np.arange(m)
        return 1.0 / (0.5 * (a + b) + 0.5 * (b - a) * np.cos(np.pi * (2 * k + 1) / (2 * m)))

    # ---- preconditioners
    def vcycle(self, l, b, exact_below=None, P0=None):
        if l == self.L - 1:
            return self.coarse_inv @ b
        if exact_below is not None and l >= exact_below:
            if l not in self.exact:
                self.exact[l] = spla.splu(sp.csc_matrix(self.A[l] + 1e-12 * sp.eye(self.A[l].shape[0]))).solve
            return self.exact[l](b)
        A, d, P = self.A[l], self.A[l].diagonal(), (P0 if (l == 0 and P0 is not None) else self.P[l])
        x = np.zeros_like(b)
        for w in self.w[l]:
            x = x + w * (b - A @ x) / d
        x = x + P @ self.vcycle(l + 1, P.T @ (b - A @ x), exact_below)
        for w in reversed(self.w[l]):
            x = x + w * (b - A @ x) / d
        return x

    def twogrid(self, P0):
        """level 0 with prolongator P0 and an exact coarse solve of P0' A P0 (prototype aggregations)"""
        A, d, om = self.A0, self.A0.diagonal(), self.w[0][0]
        Ac = (P0.T @ A @ P0).tocsc()
        lu = spla.splu(Ac + 1e-10 * sp.eye(Ac.shape[0]))

        def M(b):
            x = om * b / d
            x = x + P0 @ lu.solve(P0.T @ (b - A @ x))
            return x + om * (b - A @ x) / d
        return M

    # ---- PCG with the reference's stopping rule (Krylov.cg, src/core.jl:639), Ritz values from its coefficients
    def pcg(self, b, Minv, tol=1e-6, maxit=200):
        A = self.A0
        x = np.zeros_like(b)
        r = b.copy()
        z = Minv(r)
        p = z.copy()
        rz = r @ z
        rz0 = rz
        al, be = [], []
        for k in range(maxit):
            Ap = A @ p
            a = rz / (p @ Ap)
            x += a * p
            r -= a * Ap
            z = Minv(r)
            rzn = r @ z
            al.append(a)
            if np.sqrt(abs(rzn)) <= 1.49e-8 + tol * np.sqrt(rz0):
                return x, k + 1, al, be
            be.append(rzn / rz)
            p = z + be[-1] * p
            rz = rzn
        return x, maxit, al, be

    def run(self, name, Minv, npairs=8):
        its = []
        for a, b_ in zip(self.ids[:npairs], self.ids[8:8 + npairs]):
            b = np.zeros(self.ncell)
            b[self.n2c[b_]] = 1
            b[self.n2c[a]] = -1
            its.append(self.pcg(b, Minv)[1])
        print("%-44s iterations %s mean %.2f" % (name, its, np.mean(its)), flush=True)
        return its

    def lowspec(self, Minv, k=12, show=0, tag=""):
        """lowest eigenpairs of M^-1 A on the giant component: M^-1 y = mu A^-1 y"""
        gc = self.giant_cells
        Ag = self.A0[gc][:, gc].tocsc()
        luA = spla.splu(Ag)
        n = gc.size

        def Mi(y):
            full = np.zeros(self.ncell)
            full[gc] = np.asarray(y).ravel()
            return Minv(full)[gc]
        OP = spla.LinearOperator((n, n), matvec=Mi, dtype=np.float64)
        Mop = spla.LinearOperator((n, n), matvec=lambda y: luA.solve(np.asarray(y).ravel()), dtype=np.float64)
        Miv = spla.LinearOperator((n, n), matvec=lambda y: Ag @ np.asarray(y).ravel(), dtype=np.float64)
        mu, Y = spla.eigsh(OP, k=k, M=Mop, Minv=Miv, which="SA", ncv=4 * k, tol=1e-6)
        print(tag, "lowest eigenvalues of M^-1 A:", np.round(mu, 4), flush=True)
        out = []
        for q in range(min(show, k)):
            x = luA.solve(Y[:, q])
            w = x ** 2 / np.sum(x ** 2)
            top = gc[np.argmax(np.abs(x))]
            print("   mode %d lambda %.4f participation %.0f cells, centre (row %d, col %d)"
                  % (q, mu[q], 1 / np.sum(w ** 2), top % self.R, top // self.R))
            out.append((mu[q], x, top))
        return mu, out

    def show(self, centre, x, agg, rad=5):
        full = np.zeros(self.ncell)
        full[self.giant_cells] = x
        m = np.max(np.abs(full))
        r0, c0 = int(centre % self.R), int(centre // self.R)
        for r in range(max(0, r0 - rad), min(self.R, r0 + rad + 1)):
            s1 = s2 = ""
            for c in range(max(0, c0 - rad), min(self.C, c0 + rad + 1)):
                cell = c * self.R + r
                if not self.valid[cell]:
                    s1 += "  . "
                    s2 += "   . "
                else:
                    a = agg[cell]
                    s1 += " %s%s%s" % (chr(65 + (a % self.Rc) % 26), chr(97 + (a // self.Rc) % 26), " " if a == self.tile[cell] else "*")
                    s2 += "%5d" % int(round(100 * full[cell] / m))
            print("   r%4d" % r, s1, "   ", s2)

    # ---- aggregation prototypes (level 0, cell space)
    def aggregate(self, rule="eight"):
        """pieces of every 3x3 tile (rule: which couplings connect inside a tile), largest piece keeps the tile, the other
        cells join the tile of the main-piece cell they are most strongly coupled to, orphans adopt a neighbour's"""
        Aoff = (self.A0 - sp.diags(self.A0.diagonal())).tocoo()
        i, j = Aoff.row, Aoff.col
        keep = self.tile[i] == self.tile[j]
        diag = (np.abs(self.rows[i] - self.rows[j]) == 1) & (np.abs(self.cols[i] - self.cols[j]) == 1)
        if rule == "four":
            keep &= ~diag
        if rule == "supported":   # a diagonal link counts only next to a valid side cell
            c1, c2 = self.cols[j] * self.R + self.rows[i], self.cols[i] * self.R + self.rows[j]
            keep &= (~diag) | self.valid[c1] | self.valid[c2]
        G = sp.csr_matrix((np.ones(keep.sum()), (i[keep], j[keep])), shape=(self.ncell, self.ncell))
        ncomp, lab = csg.connected_components(G, directed=False)
        vi = np.flatnonzero(self.valid)
        key = self.tile[vi].astype(np.int64) * (ncomp + 1) + lab[vi]
        uk, cnt = np.unique(key, return_counts=True)
        tk = uk // (ncomp + 1)
        order = np.lexsort((np.arange(len(uk)), -cnt, tk))
        first = np.ones(len(uk), bool)
        first[1:] = tk[order][1:] != tk[order][:-1]
        ismain = np.zeros(self.ncell, bool)
        ismain[vi] = np.isin(key, uk[order][first])
        agg = np.full(self.ncell, -1)
        agg[ismain] = self.tile[ismain]
        A_ = Aoff.tocsr()
        for c in np.flatnonzero(self.valid & ~ismain):
            s, e = A_.indptr[c], A_.indptr[c + 1]
            nb, w = A_.indices[s:e], np.abs(A_.data[s:e])
            ok = ismain[nb] & (self.tile[nb] != self.tile[c])
            if ok.any():
                agg[c] = self.tile[nb[np.argmax(np.where(ok, w, -1))]]
        for _ in range(3):
            new = agg.copy()
            for c in np.flatnonzero(self.valid & (agg < 0)):
                s, e = A_.indptr[c], A_.indptr[c + 1]
                nb, w = A_.indices[s:e], np.abs(A_.data[s:e])
                ok = agg[nb] >= 0
                if ok.any():
                    new[c] = agg[nb[np.argmax(np.where(ok, w, -1))]]
            agg = new
        return agg

    def prolongator(self, agg, omega_p=1.6):
        """P = T - omega_p Dl^-1 A T (amg_setup.h smooth_prolongator_kernel), T_i = 1 / sqrt(size of the aggregate)"""
        nagg = self.Rc * self.Cc
        w = self.valid & (agg >= 0)
        size = np.bincount(agg[w], minlength=nagg).astype(float)
        T = sp.csr_matrix((1.0 / np.sqrt(size[agg[w]]), (np.flatnonzero(w), agg[w])), shape=(self.ncell, nagg))
        labs = np.asarray(abs(self.A0).sum(axis=1)).ravel()
        return (T - sp.diags(omega_p / labs) @ (self.A0 @ T)).tocsr()

    # ---- enrichment prototypes
    def local_graph(self, cells):
        A_ = self.A0
        idx = {c: k for k, c in enumerate(cells)}
        m = len(cells)
        W, dg = np.zeros((m, m)), np.zeros(m)
        for k, c in enumerate(cells):
            s, e = A_.indptr[c], A_.indptr[c + 1]
            for nb, v in zip(A_.indices[s:e], A_.data[s:e]):
                if nb == c:
                    dg[k] = v
                elif nb in idx:
                    W[k, idx[nb]] = -v
        return W, dg

    def enrichment(self, agg, tau, method="coord", psteps=6):
        """second coarse function of every aggregate whose local Fiedler value (L_agg phi = lambda D phi) is below tau"""
        order = np.argsort(agg[self.valid], kind="stable")
        vc, av = np.flatnonzero(self.valid)[order], agg[self.valid][order]
        starts = np.flatnonzero(np.r_[True, av[1:] != av[:-1]])
        ends = np.r_[starts[1:], len(av)]
        rows, cols, vals, k = [], [], [], 0
        for s, e in zip(starts, ends):
            cells = vc[s:e]
            if len(cells) < 3:
                continue
            W, dg = self.local_graph(cells.tolist())
            Lm = np.diag(W.sum(1)) - W
            if method == "exact":
                Dh = 1 / np.sqrt(dg)
                w, V = np.linalg.eigh(Dh[:, None] * Lm * Dh[None, :])
                phi, lam = V[:, 1] * Dh, w[1]
            else:
                r, c = (cells % self.R).astype(float), (cells // self.R).astype(float)
                cen = lambda v: v - (dg @ v) / dg.sum()    # noqa: E731
                rq = lambda v: (v @ Lm @ v) / (v @ (dg * v))   # noqa: E731
                cand = [cen(v) for v in (r, c, r + c, r - c)]
                cand = [v for v in cand if v @ v > 1e-12]
                if not cand:
                    continue
                phi = min(cand, key=rq)
                for _ in range(psteps):
                    phi = cen(phi - 0.6 * (Lm @ phi) / dg)
                lam = rq(phi)
            if lam < tau:
                rows += cells.tolist()
                cols += [k] * len(cells)
                vals += phi.tolist()
                k += 1
        return sp.csr_matrix((vals, (rows, cols)), shape=(self.ncell, k))

    def enriched(self, M0, E, B="diag", mode="mult", gamma=1.0):
        A = self.A0
        AE = (A @ E).tocsr()
        G = (E.T @ AE).tocsc()
        dG = G.diagonal()
        if mode == "add":
            return lambda r: M0(r) + gamma * (E @ ((E.T @ r) / dG))
        Bf = spla.splu(G).solve if B == "exact" else (lambda t: t / dG)

        def M(r):
            t = E.T @ r
            c = Bf(t)
            z = M0(r - AE @ c)
            return z + E @ (c + Bf(t - G @ c - AE.T @ z))
        return M


def main():
    m = Model(sys.argv[1])
    cmd = sys.argv[2] if len(sys.argv) > 2 else "iters"
    V = lambda r: m.vcycle(0, r)          # noqa: E731
    TG = lambda r: m.vcycle(0, r, exact_below=1)   # noqa: E731
    print("library:", m.d.get("library_iters"))
    if cmd == "iters":
        m.run("V-cycle (model of the library)", V)
        m.run("two-grid, exact level 1", TG)
    elif cmd == "spectrum":
        m.lowspec(TG, 16, show=16, tag="two-grid")
    elif cmd == "patterns":
        agg = m.aggregate("eight")
        _, out = m.lowspec(TG, 10, show=10, tag="two-grid")
        for lam, x, top in out[1:]:
            print("lambda %.4f" % lam)
            m.show(top, x, agg)
    elif cmd == "aggregation":
        for rule in ("eight", "supported", "four"):
            agg = m.aggregate(rule)
            m.run("two-grid, pieces by rule %s" % rule, m.twogrid(m.prolongator(agg)))
    elif cmd == "enrich":
        agg = m.aggregate("eight")
        for tau in (0.06, 0.1, 0.15):
            for method, ps in (("exact", 0), ("coord", 6)):
                E = m.enrichment(agg, tau, method, ps)
                m.run("V + enrichment tau %.2f %s (%d vectors)" % (tau, method, E.shape[1]), m.enriched(V, E))
        E = m.enrichment(agg, 0.1, "exact")
        m.run("V + ADDITIVE enrichment tau 0.10", m.enriched(V, E, mode="add"))


if __name__ == "__main__":
    main()
