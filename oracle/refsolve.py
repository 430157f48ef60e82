"""TEST INFRASTRUCTURE ONLY -- restatement of the reference's pairwise solver layer on the CPU.

Restates (Circuitscape.jl, paths relative to /root/reference):
  solve(prob, ::AMGSolver, flags, cfg, log)   src/core.jl:96-305   (pair loop, regularisation :161,
                                              RHS :224-226, grounding :231-232, result matrix :130,:294-299)
  get_num_pairs / smash_repeats!              src/core.jl:537-603
  solve_linear_system                         src/core.jl:636-643  (tolerances, residual check)
  update_voltmatrix! / update_shortcut_resistances!   src/core.jl:685-739
  _pt_file_polygons_path                      src/raster/pairwise.jl:72-135

The linear solves are executed by oracle/cs_oracle.cpp (libcs_oracle.so, the C++ restatement of
AlgebraicMultigrid.jl smoothed_aggregation + Krylov.jl cg) or, for `mode="direct"`, by a grounded sparse LU
(tight ground truth for tiny cases). Pure-Python loops: small cases only.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes
import os

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

RESISTANCE_INVALID = -777.0


class _Opts(ctypes.Structure):
    _fields_ = [("theta", ctypes.c_double), ("omega", ctypes.c_double), ("max_levels", ctypes.c_int),
                ("max_coarse", ctypes.c_int), ("improve_iters", ctypes.c_int), ("reserved", ctypes.c_int)]


class _Result(ctypes.Structure):
    _fields_ = [("iters", ctypes.c_int), ("status", ctypes.c_int), ("final_mnorm", ctypes.c_double),
                ("true_relres", ctypes.c_double), ("seconds", ctypes.c_double)]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libcs_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle library not built: run `make -C oracle`")
        L = ctypes.CDLL(path)
        L.cso_setup.restype = ctypes.c_void_p
        L.cso_setup.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                ctypes.c_int, ctypes.c_void_p]
        L.cso_free.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.cso_info.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        L.cso_solve.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                ctypes.c_void_p]
        L.cso_solve_pairs.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_void_p]
        L.cso_spmv_seconds.restype = ctypes.c_double
        L.cso_spmv_seconds.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        _LIB = L
    return _LIB


def regularize(matrix, dtype=np.float64):
    """core.jl:161  matrix.nzval .+= eps(T) * norm(matrix.nzval)  (every STORED entry is shifted)."""
    m = matrix.tocsr().astype(dtype).copy()
    # The norm is accumulated in double and rounded to T (what BLAS nrm2 kernels and the device do). numpy's float32
    # norm is a float32 dot product: on 8e7 entries it is off by ~4e-4 relative, which at precision = single moves the
    # shift -- the only grounding of that problem -- and with it every resistance by ~2e-4 (measured, round 4).
    T = np.dtype(dtype).type
    m.data = (m.data + T(np.finfo(dtype).eps) * T(np.linalg.norm(m.data.astype(np.float64)))).astype(dtype)
    return m


class OracleAMG:
    """CPU smoothed-aggregation AMG + Krylov-style PCG (the restated reference solver)."""

    def __init__(self, matrix, precision="double", theta=0.0, omega=4.0 / 3.0, max_levels=10, max_coarse=10,
                 improve_iters=4):
        m = matrix.tocsr()
        m.sort_indices()
        self.n = m.shape[0]
        self.vb = 8 if precision == "double" else 4
        self._rp = np.ascontiguousarray(m.indptr, dtype=np.int64)
        self._ci = np.ascontiguousarray(m.indices, dtype=np.int64)
        self._va = np.ascontiguousarray(m.data, dtype=np.float64)
        o = _Opts(theta, omega, max_levels, max_coarse, improve_iters, 0)
        self.h = lib().cso_setup(self.n, self._rp.ctypes.data, self._ci.ctypes.data, self._va.ctypes.data, 0, self.vb,
                                 ctypes.byref(o))
        info = np.zeros(64)
        nl = lib().cso_info(self.h, self.vb, info.ctypes.data, 64)
        self.levels = nl
        self.setup_seconds = float(info[1])
        self.operator_complexity = float(info[2])
        self.level_sizes = [(int(info[3 + 2 * l]), int(info[4 + 2 * l])) for l in range(min(nl, 30))]

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().cso_free(self.h, self.vb)
                self.h = None
        except Exception:
            pass

    def solve(self, b, rtol=1e-6, atol=-1.0, itmax=100000, criterion=0, nthreads=1):
        """b: (n,) or (n, nrhs). atol < 0 -> sqrt(eps(T)) (Krylov.jl default). Returns x, results."""
        b = np.asarray(b, dtype=np.float64)
        one = b.ndim == 1
        B = np.asfortranarray(b.reshape(self.n, -1))
        nrhs = B.shape[1]
        X = np.zeros_like(B, order="F")
        res = (_Result * nrhs)()
        lib().cso_solve(self.h, self.vb, B.ctypes.data, X.ctypes.data, nrhs, rtol, atol, itmax, criterion, nthreads,
                        ctypes.byref(res))
        out = [dict(iters=r.iters, status=r.status, final_mnorm=r.final_mnorm, true_relres=r.true_relres,
                    seconds=r.seconds) for r in res]
        return (X[:, 0] if one else X), out

    def solve_pairs(self, src, dst, gather=None, rtol=1e-6, atol=-1.0, itmax=100000, criterion=0, nthreads=1):
        """0-based local node indices. Returns resistances, gathered (npairs x ngather, grounded at src), results."""
        src = np.ascontiguousarray(src, dtype=np.int64)
        dst = np.ascontiguousarray(dst, dtype=np.int64)
        npairs = len(src)
        g = np.ascontiguousarray(gather if gather is not None else [], dtype=np.int64)
        gathered = np.zeros((npairs, len(g)))
        resist = np.zeros(npairs)
        res = (_Result * max(npairs, 1))()
        lib().cso_solve_pairs(self.h, self.vb, src.ctypes.data, dst.ctypes.data, npairs, g.ctypes.data, len(g),
                              gathered.ctypes.data, resist.ctypes.data, rtol, atol, itmax, criterion, nthreads,
                              ctypes.byref(res))
        out = [dict(iters=r.iters, status=r.status, final_mnorm=r.final_mnorm, true_relres=r.true_relres,
                    seconds=r.seconds) for r in res[:npairs]]
        return resist, gathered, out

    def spmv_seconds(self, reps=5):
        return lib().cso_spmv_seconds(self.h, self.vb, reps)


class DirectSolver:
    """Tight ground truth for small components: ground the last node, sparse LU (fp64)."""

    def __init__(self, matrix):
        m = matrix.tocsc().astype(np.float64)
        self.n = m.shape[0]
        if self.n > 1:
            self.lu = spla.splu(m[: self.n - 1, : self.n - 1].tocsc())

    def solve(self, b):
        x = np.zeros(self.n)
        if self.n > 1:
            x[: self.n - 1] = self.lu.solve(np.asarray(b, dtype=np.float64)[: self.n - 1])
        return x


def solve_linear_system(solver, matrix, b, mode):
    """core.jl:636-643 (mode 'reference': rtol 1e-6 / atol sqrt(eps) on the M-norm residual, then the 1e-4 check);
    mode 'tight': true-residual rtol 1e-12; mode 'direct': grounded LU."""
    if mode == "direct":
        return solver.solve(b)
    if mode == "tight":
        v, res = solver.solve(b, rtol=1e-12, atol=0.0, criterion=1)
    else:
        v, res = solver.solve(b)
    if not (res[0]["true_relres"] < 1e-4):
        raise RuntimeError("CG solver did not converge: relative residual %g exceeds tolerance 1e-4"
                           % res[0]["true_relres"])
    return v


def get_num_pairs(ccs, fp, exclude, user_points=None, shortcut=False):
    """core.jl:537-587."""
    user_points = fp if user_points is None else user_points
    g2u = {}
    for k in range(len(fp)):
        g2u[int(fp[k])] = int(user_points[k])  # later entries overwrite, as in Julia's Dict constructor
    exclude = set(exclude)
    num = 0
    d = {}
    for cc in ccs:
        ccset = set(int(x) for x in cc)
        sub = []
        for p in fp:
            p = int(p)
            if p in ccset and p not in sub:
                sub.append(p)
        for ii in range(len(sub)):
            if shortcut and ii > 0:
                break
            for jj in range(ii + 1, len(sub)):
                if (g2u.get(sub[ii], sub[ii]), g2u.get(sub[jj], sub[jj])) in exclude:
                    continue
                num += 1
                d[(sub[ii], sub[jj])] = num
    return num, d


def single_ground_all_pairs(prob, outputflags, mode="reference", precision="double", stats=None):
    """core.jl:96-305 for ::AMGSolver. Returns the (P+1)x(P+1) matrix with user ids in row/col 0.

    outputflags: dict with write_volt_maps, write_cur_maps, write_cum_cur_map_only, write_max_cur_maps.
    """
    dtype = np.float64 if precision == "double" else np.float32
    a = prob.G.tocsr()
    points = [int(p) for p in prob.points]
    orig_pts = [int(p) for p in prob.user_points]
    exclude = set((int(x), int(y)) for x, y in prob.exclude_pairs)
    numpoints = len(points)
    resistances = -np.ones((numpoints, numpoints))
    voltmatrix = np.zeros((numpoints, numpoints))
    shortcut_res = -np.ones((numpoints, numpoints))
    shortcut = (prob.is_raster and not outputflags.get("write_volt_maps") and not outputflags.get("write_cur_maps")
                and not outputflags.get("write_cum_cur_map_only") and not outputflags.get("write_max_cur_maps")
                and len(exclude) == 0)
    nsolves = 0
    for comp in prob.cc:
        compset = {int(x): k for k, x in enumerate(comp)}
        csub = []
        for p in points:
            if p in compset and p not in csub:
                csub.append(p)
        if not csub:
            continue
        idx0 = np.asarray(comp, dtype=np.int64) - 1
        matrix = regularize(a[idx0][:, idx0], dtype)
        solver = DirectSolver(matrix) if mode == "direct" else OracleAMG(matrix, precision)
        n = matrix.shape[0]

        def solve_pairs_for_point(point_idx):
            nonlocal nsolves
            results = []
            src_node = csub[point_idx]
            comp_i = compset[src_node]
            src_indices = [k for k, p in enumerate(points) if p == src_node]
            for x in range(len(src_indices)):           # smash_repeats!
                for y in range(x + 1, len(src_indices)):
                    results.append((src_indices[x], src_indices[y], 0.0))
            for pair_idx in range(point_idx + 1, len(csub)):
                dst_node = csub[pair_idx]
                comp_j = compset[dst_node]
                dst_indices = [k for k, p in enumerate(points) if p == dst_node]
                if src_node == dst_node:
                    continue
                if not any((orig_pts[ci], orig_pts[cj]) not in exclude for ci in src_indices for cj in dst_indices):
                    continue
                current = np.zeros(n)
                current[comp_i] = -1.0
                current[comp_j] = 1.0
                voltages = solve_linear_system(solver, matrix, current, mode)
                nsolves += 1
                voltages = voltages - voltages[comp_i]
                resistance = voltages[comp_j] - voltages[comp_i]
                for ci in src_indices:
                    for cj in dst_indices:
                        if (orig_pts[ci], orig_pts[cj]) in exclude:
                            continue
                        results.append((ci, cj, resistance))
                        if shortcut:
                            resistances[ci, cj] = resistance
                            resistances[cj, ci] = resistance
                            # update_voltmatrix! (core.jl:685-703)
                            for i in range(1, numpoints):
                                ind = compset.get(points[i])
                                if ind is not None:
                                    voltmatrix[i, cj] = 1.0 - voltages[ind] / resistance
            return results

        if shortcut:
            anchor = points.index(csub[0])
            solve_pairs_for_point(0)
            _update_shortcut_resistances(anchor, voltmatrix, shortcut_res, resistances, points, compset)
        else:
            for pt in range(len(csub)):
                for ci, cj, rv in solve_pairs_for_point(pt):
                    resistances[ci, cj] = rv
                    resistances[cj, ci] = rv
    if shortcut:
        resistances = shortcut_res
    for i in range(numpoints):
        resistances[i, i] = 0.0
    r = np.zeros((numpoints + 1, numpoints + 1))
    r[0, 1:] = orig_pts
    r[1:, 0] = orig_pts
    r[1:, 1:] = resistances
    if stats is not None:
        stats["nsolves"] = stats.get("nsolves", 0) + nsolves
        stats["shortcut"] = shortcut
    return r


def _update_shortcut_resistances(anchor, voltmatrix, shortcut, resistances, points, compset):
    """core.jl:706-739."""
    l = resistances.shape[0]
    check = [p in compset for p in points]
    for pointx in range(l):
        if not check[pointx]:
            continue
        R1x = resistances[anchor, pointx]
        if R1x == -1:
            continue
        shortcut[pointx, anchor] = shortcut[anchor, pointx] = R1x
        for point2 in range(pointx, l):
            if not check[point2]:
                continue
            R12 = resistances[anchor, point2]
            if R12 == -1:
                continue
            if R1x != RESISTANCE_INVALID:
                shortcut[anchor, point2] = shortcut[point2, anchor] = R12
                Vx = voltmatrix[pointx, point2]
                R2x = 2 * R12 * Vx + R1x - R12
                if shortcut[point2, pointx] != RESISTANCE_INVALID:
                    shortcut[point2, pointx] = shortcut[pointx, point2] = R2x
            else:
                shortcut[pointx, :] = RESISTANCE_INVALID
                shortcut[:, pointx] = RESISTANCE_INVALID


def raster_pairwise_from_fixture(case, mode="reference", precision="double", stats=None):
    """raster_pairwise (raster/pairwise.jl:14-30) on a tests/golden fixture dict."""
    from . import refgraph as rg
    o = case["options"]
    gmap = np.array(case["cellmap"], dtype=np.float64)
    polymap = np.array(case["polymap"], dtype=np.int64) if case["polymap"] is not None else None
    points_rc = tuple(list(x) for x in case["points_rc"])
    flags = {k: o[k] for k in ("write_volt_maps", "write_cur_maps", "write_cum_cur_map_only", "write_max_cur_maps")}
    avg_res = o["connect_using_avg_resistances"]
    four = o["connect_four_neighbors_only"]
    contains_polygons = len(points_rc[0]) != len(set(points_rc[2]))
    if not contains_polygons:
        prob = rg.compute_graph_data_no_polygons(gmap, polymap, points_rc, case["included_pairs"], avg_res, four)
        return single_ground_all_pairs(prob, flags, mode, precision, stats)
    # _pt_file_polygons_path (raster/pairwise.jl:72-135)
    if case["included_pairs"] is not None:
        exclude, points_rc = rg.generate_exclude_pairs(points_rc, case["included_pairs"])
    else:
        exclude = []
    exclude = set(exclude)
    pts = []
    for v in points_rc[2]:
        if v not in pts:
            pts.append(v)
    res = -np.ones((len(pts), len(pts)))
    for i in range(len(pts)):
        for j in range(i + 1, len(pts)):
            if (pts[i], pts[j]) in exclude or (pts[j], pts[i]) in exclude:
                continue
            prob = rg.compute_graph_data_polygons(gmap, polymap, points_rc, pts[i], pts[j], avg_res, four)
            pr = single_ground_all_pairs(prob, flags, mode, precision, stats)
            res[i, j] = res[j, i] = pr[1, 2]
    for i in range(len(pts)):
        res[i, i] = 0
    r = np.zeros((len(pts) + 1, len(pts) + 1))
    r[0, 1:] = pts
    r[1:, 0] = pts
    r[1:, 1:] = res
    return r


def network_pairwise_from_fixture(case, mode="reference", precision="double", stats=None):
    from . import refgraph as rg
    prob = rg.compute_graph_data_network(case["edges_i"], case["edges_j"], case["edges_v"], case["focal"])
    flags = dict(write_volt_maps=False, write_cur_maps=False, write_cum_cur_map_only=False, write_max_cur_maps=False)
    return single_ground_all_pairs(prob, flags, mode, precision, stats)


# ---------------------------------------------------------------------------------------------------------------
# advanced mode, network flavour (scope row N2): restates
#   _get_sources_and_grounds / resolve_conflicts   src/raster/advanced.jl:84-149
#   advanced_kernel (network branch, no maps)      src/raster/advanced.jl:151-271
#   multiple_solver / multiple_solve(::AMGSolver)  src/raster/advanced.jl:274-312
def resolve_conflicts(sources, grounds, policy):
    sources = np.array(sources, dtype=np.float64)
    grounds = np.array(grounds, dtype=np.float64)
    finitegrounds = np.where(grounds < np.inf, grounds, 0.0)
    if np.count_nonzero(finitegrounds) == 0:
        finitegrounds = np.array([-9999.0])
    conflicts = (sources != 0) & (grounds != 0)
    if conflicts.any():
        if policy in ("rmvsrc", "rmvall"):
            sources[conflicts] = 0
        elif policy == "rmvgnd":
            grounds[conflicts] = 0
    infgrounds = grounds == np.inf
    infconflicts = infgrounds & (sources > 0)
    grounds[infconflicts] = 0
    return sources, grounds, finitegrounds


def multiple_solver(a, sources, grounds, finitegrounds, solve):
    """advanced.jl:274-305. `solve(matrix, rhs) -> x` is the multiple_solve hook."""
    asolve = sp.csr_matrix(a, dtype=np.float64)
    if not (len(finitegrounds) == 1 and finitegrounds[0] == -9999):
        asolve = (asolve + sp.diags(finitegrounds)).tocsr()
    inf = np.flatnonzero(grounds == np.inf)
    keep = np.setdiff1d(np.arange(a.shape[0]), inf)
    asolve = asolve[keep][:, keep]
    volt = solve(asolve, np.asarray(sources, dtype=np.float64)[keep])
    voltages = np.zeros(a.shape[0])
    voltages[keep] = volt
    return voltages


def _oracle_multiple_solve(mode):
    def solve(matrix, rhs):
        if mode == "direct":
            return spla.spsolve(matrix.tocsc(), rhs)
        S = OracleAMG(matrix)  # smoothed_aggregation(matrix) with all defaults == the pairwise setup (advanced.jl:308)
        v = solve_linear_system(S, matrix, rhs, "tight" if mode == "tight" else "reference")
        return v
    return solve


def network_advanced_from_fixture(case, mode="reference", solve=None):
    """network_advanced (network/advanced.jl:1-17) on a tests/golden fixture: returns [node id (1-based), voltage]."""
    from . import refgraph as rg
    ei = np.asarray(case["edges_i"]); ej = np.asarray(case["edges_j"])
    m = int(max(ei.max(), ej.max()))
    A = sp.coo_matrix((np.asarray(case["edges_v"], dtype=np.float64), (ei - 1, ej - 1)), shape=(m, m)).tocsr()
    A = (A + A.T).tocsr()
    cc = rg.connected_components(A)
    G = rg.laplacian(A)
    sources = np.zeros(m); grounds = np.zeros(m)
    gl = np.array(case["grounds"], dtype=np.float64)
    if case["ground_file_is_resistances"]:
        with np.errstate(divide="ignore"):
            gl[:, 1] = 1.0 / gl[:, 1]
    for node, val in case["sources"]:
        sources[int(node) - 1] = val
    for node, val in gl:
        grounds[int(node) - 1] = val
    sources, grounds, finitegrounds = resolve_conflicts(sources, grounds, case["remove_src_or_gnd"])
    solve = solve or _oracle_multiple_solve(mode)
    voltages = np.zeros(m)
    for c in cc:
        idx = np.asarray(c) - 1
        s_local, g_local = sources[idx], grounds[idx]
        if s_local.sum() == 0 or g_local.sum() == 0:
            continue
        f_local = finitegrounds[idx] if not (len(finitegrounds) == 1 and finitegrounds[0] == -9999) else finitegrounds
        voltages[idx] += multiple_solver(G[idx][:, idx], s_local, g_local, f_local, solve)
    return np.column_stack([np.arange(1, m + 1), voltages])
