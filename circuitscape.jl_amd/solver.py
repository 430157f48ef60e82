"""Host-side mirror of the reference's solver-layer interface for the HIP backend.

Julia is not available in the build image, so the host code above the C ABI is written in Python and mirrors the
reference's own operator / plug-in API for this path -- same names, argument meaning and error behaviour
(Circuitscape.jl, paths relative to /root/reference):

    HIPAMGSolver(bs)                         <-> struct CholmodSolver/PardisoSolver{bs}          src/core.jl:48-63
    get_solver(cfg)                          <-> get_solver(cfg)                                 src/core.jl:74-94
    construct_cholesky_factor(matrix, s)     <-> construct_cholesky_factor(matrix, ::XSolver)    src/core.jl:519-523,
                                                                                                 ext/CircuitscapePardisoExt.jl:31-32
    solve_linear_system(factor, matrix, rhs) <-> solve_linear_system(factor, matrix, rhs)        src/core.jl:646-653
    multiple_solve(s, matrix, sources)       <-> multiple_solve(s::XSolver, matrix, sources)     src/raster/advanced.jl:314-333
    solve(prob, s, flags, cfg, log)          <-> solve(prob, ::AMGSolver / ::Union{...}, ...)    src/core.jl:96-305, 312-515
    single_ground_all_pairs(prob, flags, cfg)<-> single_ground_all_pairs                         src/core.jl:70-72

The equivalent Julia glue (package extension + the ~15-line patch to consts.jl/config.jl/core.jl) is shown in
INTEGRATION.md. Everything numerical happens in libcsgpu.so (hand-written HIP kernels); this module only does the
integer bookkeeping of the pair loop. There is no CPU fallback: `lib` raises if the HIP library is missing.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sp

from . import lib

RESISTANCE_INVALID = -777.0  # src/consts.jl:45

# solver aliases a patched consts.jl would carry next to AMG/CHOLMOD/PARDISO/ACCELERATE (src/consts.jl:12-15)
HIP = ["hip", "hip+amg", "cg+amg+hip", "mi355x"]


@dataclass
class HIPAMGSolver:
    """AMG-preconditioned CG on the MI355X; `bs` = right-hand sides per SpMM batch (cf. CholmodSolver.bs)."""
    bs: int = 8
    opts: dict = field(default_factory=dict)  # extra csgpu_opts overrides (rtol, criterion, theta, ...)


@dataclass
class OutputFlags:  # src/out.jl:1-10
    write_volt_maps: bool = False
    write_cur_maps: bool = False
    write_cum_cur_map_only: bool = False
    write_max_cur_maps: bool = False
    set_null_currents_to_nodata: bool = False
    set_null_voltages_to_nodata: bool = False
    log_transform_maps: bool = False


@dataclass
class Cumulative:
    """src/core.jl:1-8 (raster part): cumulative and maximum current maps shared by all pairs of a job."""
    cum_curr: np.ndarray
    max_curr: Optional[np.ndarray] = None


def initialize_cum_maps(cellmap, write_max=False):
    """src/utils.jl:122-131."""
    cellmap = np.asarray(cellmap)
    return Cumulative(np.zeros(cellmap.shape), np.full(cellmap.shape, -9999.0) if write_max else None)


@dataclass
class Flags:  # the subset of RasterFlags / NetworkFlags the solver layer reads (raster/pairwise.jl:1-12)
    is_raster: bool = True
    outputflags: OutputFlags = field(default_factory=OutputFlags)
    is_onetoall: bool = False
    is_alltoone: bool = False
    policy: str = "keepall"          # remove_src_or_gnd: keepall | rmvsrc | rmvgnd | rmvall


@dataclass
class GraphProblem:
    """src/core.jl:10-22. Node ids (`points`, `cc`, `nodemap`) are 1-based as in the reference, 0 = no node."""
    G: sp.csr_matrix
    cc: List[np.ndarray]
    points: np.ndarray
    user_points: np.ndarray
    exclude_pairs: Sequence[Tuple[int, int]] = ()
    nodemap: Optional[np.ndarray] = None
    polymap: Optional[np.ndarray] = None
    solver: HIPAMGSolver = field(default_factory=HIPAMGSolver)
    cellmap: Optional[np.ndarray] = None     # conductance raster (for the set_null_*_to_nodata options)
    cum: Optional[Cumulative] = None         # cumulative / maximum current maps (scope row N1)
    net_coords: Optional[Sequence[Tuple[int, int]]] = None   # network mode: edge list (i, j) of the input file, for the
                                                             # cumulative branch currents (utils.jl:133-142)


def get_solver(cfg):
    """cfg: mapping with 'solver' (alias string) and optional 'cholmod_batch_size' (src/config.jl:25-29)."""
    s = cfg.get("solver", "hip")
    if s in HIP:
        bs = int(cfg.get("cholmod_batch_size", 8))
        return HIPAMGSolver(bs=min(max(bs, 1), 16))
    raise ValueError("Unknown solver: %s" % s)  # core.jl:92


def _opts_for(solver, **extra):
    kw = dict(batch=max(1, min(16, int(solver.bs))))
    kw.update(solver.opts)
    kw.update(extra)
    return lib.default_opts(**kw)


def regularize(matrix):
    """core.jl:161: matrix.nzval .+= eps(T) * norm(matrix.nzval) on the component copy (every stored entry)."""
    m = matrix.tocsr().copy()
    dt = m.dtype.type
    m.data = (m.data + np.finfo(dt).eps * dt(np.linalg.norm(m.data))).astype(dt)
    return m


def construct_cholesky_factor(matrix, solver, node_row=None, node_col=None):
    """Setup handle = device-resident matrix + AMG hierarchy (the name is the reference's plug-in hook)."""
    return lib.setup(matrix, _opts_for(solver), node_row=node_row, node_col=node_col)


def _raise_not_converged(e):
    raise RuntimeError(str(e))


def solve_linear_system(factor, matrix, rhs):
    """rhs: (n,) or (n, bs). Returns lhs of the same shape; raises with the reference's wording if a column fails the
    1e-4 true-residual check (core.jl:640-641, 649-650) -- the check itself is evaluated on the device."""
    try:
        x, _ = factor.solve_rhs(rhs)
    except lib.CsgpuError as e:
        if e.code == lib.CSGPU_NOT_CONVERGED:
            _raise_not_converged(e)
        raise
    return x


def multiple_solve(solver, matrix, sources):
    """raster/advanced.jl:307-312: fresh setup + one general right-hand side."""
    with construct_cholesky_factor(matrix, solver) as factor:
        volt = solve_linear_system(factor, matrix, np.asarray(sources))
    return volt


def _edge_weight(x, y, diag, avg_res):
    """construct_graph's edge conductance (raster/pairwise.jl:356-367): mean conductance or mean resistance, /sqrt(2)
    on diagonals."""
    with np.errstate(divide="ignore"):
        w = 1.0 / ((1.0 / x + 1.0 / y) / 2.0) if avg_res else (x + y) / 2.0
    return w / np.sqrt(2.0) if diag else w


def raster_advanced_on_device(cellmap, source_map, ground_map, flags, solver, four_neighbors=False, avg_res=False):
    """Raster advanced mode WITHOUT polygons with graph layer, solve and current map on the device (scope rows N2 + N4):
    the mirror of compute_advanced_data + advanced_kernel (src/raster/advanced.jl:36-271) for node == cell.

    * conflict policy (resolve_conflicts, advanced.jl:118-149) applied to the rasters;
    * a direct (infinite) ground -- whose row the reference deletes (multiple_solver, advanced.jl:282-288) -- becomes a
      NODATA cell, and the conductance of every edge into it is added to the ground conductance of the neighbour at the
      other end: exactly the matrix the deletion leaves behind;
    * csgpu_raster_setup_grounded + csgpu_solve_raster do the rest: all components that hold a source and a ground
      in one PCG, voltages and node currents (ground currents included) back as rasters;
    * the current into each direct ground (its own node current in the reference) is the sum of what its neighbours
      send it, evaluated here from the voltage raster for those few cells.
    Returns (voltmap, curmap) post-processed like write_grid does."""
    gmap = np.asarray(cellmap, dtype=np.float64)
    src = np.where(gmap > 0, np.asarray(source_map, dtype=np.float64), 0.0)
    gnd = np.where(gmap > 0, np.asarray(ground_map, dtype=np.float64), 0.0)
    conflicts = (src != 0) & (gnd != 0)
    if flags.policy in ("rmvsrc", "rmvall"):
        src[conflicts] = 0
    elif flags.policy == "rmvgnd":
        gnd[conflicts] = 0
    gnd[(gnd == np.inf) & (src > 0)] = 0          # a source on an infinite ground wins (advanced.jl:144-146)
    direct = gnd == np.inf
    cond = np.where(direct, 0.0, gmap)
    leak = np.where(direct, 0.0, gnd)
    R, C = gmap.shape
    nbrs = [(-1, 0), (1, 0), (0, -1), (0, 1)] + ([] if four_neighbors else [(-1, -1), (-1, 1), (1, -1), (1, 1)])
    di, dj = np.nonzero(direct)
    links = []                                     # (direct cell, neighbour cell, edge conductance)
    for i, j in zip(di, dj):
        for a, b in nbrs:
            ii, jj = i + a, j + b
            if 0 <= ii < R and 0 <= jj < C and cond[ii, jj] > 0:
                w = _edge_weight(gmap[i, j], gmap[ii, jj], a != 0 and b != 0, avg_res)
                leak[ii, jj] += w
                links.append((i, j, ii, jj, w))
    of = flags.outputflags
    try:
        with lib.raster_setup(cond, _opts_for(solver, batch=1), four_neighbors=four_neighbors, avg_resistances=avg_res,
                              reg=False, ground=leak) as h:
            cur, vol, _ = h.solve_raster(src, want_currents=True, want_voltages=True)
    except lib.CsgpuError as e:
        if e.code == lib.CSGPU_NOT_CONVERGED:
            _raise_not_converged(e)
        raise
    inflow = np.zeros(gmap.shape)
    outflow = np.zeros(gmap.shape)
    for i, j, ii, jj, w in links:                  # the direct ground sits at voltage 0
        f = w * vol[ii, jj]
        if f > 0:
            inflow[i, j] += f
        else:
            outflow[i, j] -= f
    cur = np.where(direct, np.maximum(inflow, outflow), cur)
    return (_process_grid(vol, gmap, False, of.set_null_voltages_to_nodata),
            _process_grid(cur, gmap, of.log_transform_maps, of.set_null_currents_to_nodata))


def onetoall_on_device(cellmap, points_rc, flags, solver, four_neighbors=False, avg_res=False, stats=None):
    """One-to-all / all-to-one (src/raster/onetoall.jl:13-162) for a raster without polygons and with single-cell focal
    points, with ONE graph build and ONE AMG setup for all focal points (scope rows N2 + N4): per focal point n,
    one-to-all injects unit current at n and ties every other focal cell directly to ground; all-to-one grounds n and
    injects unit current at every other focal cell. The reference deletes the grounded rows / columns and factorises
    again per point (advanced.jl:282-288, 307-312); here the points are columns of csgpu_solve_grounded -- same reduced
    systems, batches of `solver.bs` points per PCG, hierarchy of the ungrounded Laplacian -- and the node currents come
    from the device as well. Components without a source or without a ground are skipped like advanced_kernel does
    (advanced.jl:186-191). Returns (res, cum, points) like the reference's onetoall_kernel."""
    gmap = np.asarray(cellmap, dtype=np.float64)
    rows = np.asarray(points_rc[0], dtype=np.int64) - 1
    cols = np.asarray(points_rc[1], dtype=np.int64) - 1
    ids = [int(v) for v in points_rc[2]]
    assert len(ids) == len(set(ids)), "single-cell focal points only (regions need polygon merging on the host path)"
    of = flags.outputflags
    res = np.zeros(len(ids))
    cum = initialize_cum_maps(gmap, of.write_max_cur_maps)
    per_point = {}
    want_cur = of.write_cur_maps or of.write_cum_cur_map_only
    try:
        with lib.raster_setup(gmap, _opts_for(solver), four_neighbors=four_neighbors, avg_resistances=avg_res,
                              reg=False) as h:
            nodemap = h.raster_nodemap()
            n = h.info["n"]
            node = nodemap[rows, cols].astype(np.int64) - 1          # -1: the focal cell is NODATA
            comp, _ = h.components()
            sources, grounds, solvable, ground_comps = [], [], [], []
            for i in range(len(ids)):
                others = [int(node[k]) for k in range(len(ids)) if k != i and node[k] >= 0]
                if flags.is_onetoall:
                    src, gnd = ([int(node[i])] if node[i] >= 0 else []), others
                else:
                    src, gnd = others, ([int(node[i])] if node[i] >= 0 else [])
                gcomps = set(int(comp[g]) for g in gnd)
                src = [q for q in src if int(comp[q]) in gcomps]      # a component without a ground is not solved
                sources.append(src)
                grounds.append(gnd)
                solvable.append(len(src) > 0)
                ground_comps.append(gcomps)
            if not of.write_volt_maps and not (of.write_cur_maps and not of.write_cum_cur_map_only) \
                    and not of.log_transform_maps and len(ids) > 1:
                # Nothing per focal point is written: the driver keeps `res[i] = v[1]` (onetoall.jl:141) and the accumulated
                # current maps (onetoall.jl:153-158) -- csgpu_solve_sources returns exactly those: sparse right-hand sides in,
                # one voltage per point + at most two n-vectors out, the maps accumulated on the device (round 6)
                node_cum = np.zeros(n) if want_cur else None
                node_max = np.zeros(n) if (want_cur and of.write_max_cur_maps) else None
                chk = [int(node[i]) if flags.is_onetoall else -1 for i in range(len(ids))]
                v, _, _, st = h.solve_sources(sources, grounds, check=chk, cum=node_cum, mx=node_max)
                if stats is not None:
                    stats.update(st)
                for i in range(len(ids)):
                    if flags.is_onetoall:
                        res[i] = v[i] if (solvable[i] and v[i] != 0) else -1
                    else:
                        res[i] = 0 if solvable[i] else -1
                if want_cur:
                    cum.cum_curr = _process_grid(cum.cum_curr + _scatter(node_cum, nodemap), gmap, False,
                                                 of.set_null_currents_to_nodata)
                    if of.write_max_cur_maps:
                        cum.max_curr = _process_grid(np.maximum(cum.max_curr, _scatter(node_max, nodemap)), gmap, False,
                                                     of.set_null_currents_to_nodata)
                return np.column_stack([np.asarray(ids, dtype=np.float64), res]), cum, {nid: {} for nid in ids}
            B = np.zeros((n, len(ids)))
            for i, src in enumerate(sources):
                B[src, i] = 1.0
            X, C, st = h.solve_grounded(B, grounds, want_currents=want_cur)
            # a component without a ground is not part of the column's system (advanced.jl:186-191). An island whose cells
            # share a 3x3 aggregate with a solved component can pick up a constant from the preconditioner (residual-free:
            # a constant is in the kernel of its block); the reference has no voltage there
            for i, gcomps in enumerate(ground_comps):
                outside = ~np.isin(comp, list(gcomps))
                X[outside, i] = 0.0
                if C is not None:
                    C[outside, i] = 0.0
            if stats is not None:
                stats.update(st)
    except lib.CsgpuError as e:
        if e.code == lib.CSGPU_NOT_CONVERGED:
            _raise_not_converged(e)
        raise
    for i, nid in enumerate(ids):
        if len(ids) == 1:
            res[i] = -1
            continue
        vol = _scatter(X[:, i], nodemap)
        if flags.is_onetoall:
            v = vol[rows[i], cols[i]]
            res[i] = v if (solvable[i] and v != 0) else -1
        else:
            res[i] = 0 if solvable[i] else -1
        maps = {}
        if of.write_volt_maps:
            maps["voltmap"] = _process_grid(vol, gmap, False, of.set_null_voltages_to_nodata)
        if want_cur:
            cur = _scatter(C[:, i], nodemap)
            if of.write_cur_maps:
                maps["curmap"] = _process_grid(cur, gmap, of.log_transform_maps, of.set_null_currents_to_nodata)
            cum.cum_curr += cur
            if of.write_max_cur_maps:
                cum.max_curr = np.maximum(cum.max_curr, cur)
        per_point[nid] = maps
    if want_cur:
        cum.cum_curr = _process_grid(cum.cum_curr, gmap, of.log_transform_maps, of.set_null_currents_to_nodata)
        if of.write_max_cur_maps:
            cum.max_curr = _process_grid(cum.max_curr, gmap, of.log_transform_maps, of.set_null_currents_to_nodata)
    return np.column_stack([np.asarray(ids, dtype=np.float64), res]), cum, per_point


def compute_omniscape_current_batch(windows, cs_cfg, solver=None, want_voltages=False):
    """Many moving-window solves of compute_omniscape_current (src/utils.jl:145-257) as ONE device job (scope row N3).

    windows: list of (conductance, source, ground) rasters (NODATA = conductance 0, ground = finite conductances to
    ground, as Omniscape passes them). The windows are stacked into one tall raster separated by NODATA rows; the
    device graph layer (csgpu_raster_setup_grounded) numbers the cells, writes the block-diagonal Laplacian with the
    ground conductances on its diagonal and sets up ONE hierarchy; csgpu_solve_raster builds the right-hand side from
    the stacked source raster (policy :rmvsrc, components without source or ground skipped as in advanced_kernel),
    runs ONE PCG over all windows -- each window is a connected component, or several, of the same SPD system -- and
    returns the node-current raster, which is cut back into per-window maps. Nothing n-sized is built on the host.
    Returns the list of current maps (raw accumulated currents, like the reference's `outcurr`)."""
    solver = solver or get_solver({"solver": "hip", "cholmod_batch_size": cs_cfg.get("cholmod_batch_size", 8)})
    four = str(cs_cfg.get("connect_four_neighbors_only", "False")).lower() in ("true", "1")
    shapes = [np.asarray(w[0]).shape for w in windows]
    width = max(sh[1] for sh in shapes)
    height = sum(sh[0] for sh in shapes) + len(shapes) - 1
    stack = [np.zeros((height, width)) for _ in range(3)]
    offs = []
    r0 = 0
    for (cond, src, gnd), (hh, ww) in zip(windows, shapes):
        valid = np.asarray(cond, dtype=np.float64) > 0
        stack[0][r0:r0 + hh, :ww] = np.where(valid, cond, 0.0)
        gnd = np.where(valid, gnd, 0.0)
        stack[1][r0:r0 + hh, :ww] = np.where(valid & (gnd == 0), src, 0.0)   # policy :rmvsrc (utils.jl:193-196)
        stack[2][r0:r0 + hh, :ww] = gnd
        offs.append(r0)
        r0 += hh + 1                                      # one NODATA row between windows
    try:
        with lib.raster_setup(stack[0], _opts_for(solver, batch=1), four_neighbors=four, avg_resistances=False,
                              reg=False, ground=stack[2]) as h:
            cur, vol, st = h.solve_raster(stack[1], want_currents=True, want_voltages=want_voltages)
    except lib.CsgpuError as e:
        if e.code == lib.CSGPU_NOT_CONVERGED:
            _raise_not_converged(e)
        raise
    maps = [cur[o:o + hh, :ww].copy() for o, (hh, ww) in zip(offs, shapes)]
    if want_voltages:
        return maps, st, [vol[o:o + hh, :ww].copy() for o, (hh, ww) in zip(offs, shapes)]
    return maps, st


def omniscape_windows(conductance, source_strength, radius, block_size=1):
    """The moving windows of an Omniscape run, one per target block (Omniscape.jl's loop around the entry point
    compute_omniscape_current, src/utils.jl:145-257; the loop itself lives in Omniscape.jl, McRae et al. 2016): targets are
    the centres of the block_size x block_size blocks (block_size odd) whose summed source strength is positive and whose
    centre cell has positive conductance; a window is the disc of `radius` cells around the centre, clipped at the
    raster's edges. Per window: conductance (0 outside the disc), sources = the disc's source strengths with the
    target's own block zeroed, rescaled so that the injected current equals the block's summed strength, and the centre
    cell tied directly to ground (ground = inf). Yields (r0, c0, conductance, source, ground, disc) with r0, c0 the
    window's origin in the raster."""
    cond = np.asarray(conductance, dtype=np.float64)
    strength = np.asarray(source_strength, dtype=np.float64)
    assert block_size % 2 == 1 and cond.shape == strength.shape
    R, C = cond.shape
    half = block_size // 2
    for ci in range(half, R, block_size):
        for cj in range(half, C, block_size):
            b0, b1, d0, d1 = max(ci - half, 0), min(ci + half + 1, R), max(cj - half, 0), min(cj + half + 1, C)
            weight = float(np.sum(np.where(cond[b0:b1, d0:d1] > 0, strength[b0:b1, d0:d1], 0.0)))
            if not (weight > 0 and cond[ci, cj] > 0):
                continue
            r0, r1, c0, c1 = max(ci - radius, 0), min(ci + radius + 1, R), max(cj - radius, 0), min(cj + radius + 1, C)
            ii, jj = np.mgrid[r0:r1, c0:c1]
            disc = (ii - ci) ** 2 + (jj - cj) ** 2 <= radius * radius
            wc = np.where(disc, cond[r0:r1, c0:c1], 0.0)
            ws = np.where(disc & (wc > 0), strength[r0:r1, c0:c1], 0.0)
            # the target's own block, clipped to the window: with radius < block_size // 2 the block reaches beyond the
            # disc's bounding box and b0 - r0 would be negative (numpy reads that from the END of the axis: ADVICE r3)
            ws[max(b0, r0) - r0:min(b1, r1) - r0, max(d0, c0) - c0:min(d1, c1) - c0] = 0.0
            total = ws.sum()
            if not total > 0:
                continue
            ws *= weight / total
            wg = np.zeros(wc.shape)
            wg[ci - r0, cj - c0] = np.inf
            yield r0, c0, wc, ws, wg, disc


def omniscape_moving_window(conductance, source_strength, radius, block_size=1, cs_cfg=None, solver=None,
                            windows_per_solve=64):
    """Cumulative current map of an Omniscape run with every window solved on the device (scope row N3): the windows of
    omniscape_windows go down `windows_per_solve` at a time as ONE block-diagonal system (compute_omniscape_current_batch:
    one graph build, one hierarchy, one PCG per chunk) and their current maps are added into the mosaic. The directly
    grounded centre of a window becomes a NODATA cell whose edge conductances move onto its neighbours' ground
    conductances -- the matrix the reference's row deletion leaves (multiple_solver, raster/advanced.jl:282-288) -- and
    its own node current, the sum of what its neighbours send it, is evaluated from the voltage map for that one cell.
    Returns (cumulative current map, number of windows solved)."""
    cs_cfg = cs_cfg or {}
    four = str(cs_cfg.get("connect_four_neighbors_only", "False")).lower() in ("true", "1")
    nbrs = [(-1, 0), (1, 0), (0, -1), (0, 1)] + ([] if four else [(-1, -1), (-1, 1), (1, -1), (1, 1)])
    cum = np.zeros(np.asarray(conductance).shape)
    chunk, nsolved = [], 0

    def flush():
        nonlocal nsolved
        if not chunk:
            return
        maps, _, volts = compute_omniscape_current_batch([(w[2], w[3], w[4]) for w in chunk], cs_cfg, solver=solver,
                                                         want_voltages=True)
        for (r0, c0, _, _, _, ti, tj, links), cur, vol in zip(chunk, maps, volts):
            cur[ti, tj] = sum(w * vol[a, b] for a, b, w in links)     # everything the window injects sinks here
            cum[r0:r0 + cur.shape[0], c0:c0 + cur.shape[1]] += cur
        nsolved += len(chunk)
        chunk.clear()

    for r0, c0, wc, ws, wg, _ in omniscape_windows(conductance, source_strength, radius, block_size):
        ti, tj = [int(v[0]) for v in np.nonzero(wg == np.inf)]
        cond = wc.copy()
        leak = np.zeros(wc.shape)
        links = []
        for a, b in nbrs:
            ii, jj = ti + a, tj + b
            if 0 <= ii < wc.shape[0] and 0 <= jj < wc.shape[1] and wc[ii, jj] > 0:
                w = _edge_weight(wc[ti, tj], wc[ii, jj], a != 0 and b != 0, False)
                leak[ii, jj] += w
                links.append((ii, jj, w))
        cond[ti, tj] = 0.0
        chunk.append((r0, c0, cond, ws, leak, ti, tj, links))
        if len(chunk) >= windows_per_solve:
            flush()
    flush()
    return cum, nsolved


def _colmajor_nonzero(mask):
    jj, ii = np.nonzero(np.asarray(mask).T)
    return ii, jj


def _construct_node_map(gmap, polymap):
    """src/raster/pairwise.jl:271-314 (needed by construct_local_node_map when polygons are present)."""
    gmap = np.asarray(gmap, dtype=np.float64)
    nodemap = np.zeros(gmap.shape, dtype=np.int64)
    ii, jj = _colmajor_nonzero(gmap > 0)
    nodemap[ii, jj] = np.arange(1, len(ii) + 1)
    if polymap is None or np.size(polymap) == 0:
        return nodemap
    polymap = np.asarray(polymap, dtype=np.int64)
    pruned = np.where(gmap > 0, polymap, 0)
    for polynum in np.unique(polymap):
        if polynum == 0:
            continue
        i1, j1 = _colmajor_nonzero(pruned == polynum)
        if len(i1) > 0:
            nodemap[polymap == polynum] = nodemap[i1[0], j1[0]]
    ii, jj = _colmajor_nonzero(nodemap != 0)
    _, inv = np.unique(nodemap[ii, jj], return_inverse=True)
    nodemap[ii, jj] = inv + 1
    return nodemap


def construct_local_node_map(nodemap, component, polymap):
    """src/utils.jl:10-30: node map restricted to one connected component, renumbered 1..n_component."""
    nodemap = np.asarray(nodemap)
    local = np.where(np.isin(nodemap, component), nodemap, 0)
    if np.array_equal(local, nodemap):
        return local
    if polymap is None or np.size(polymap) == 0:
        ii, jj = _colmajor_nonzero(local != 0)
        local = local.copy()
        local[ii, jj] = np.arange(1, len(ii) + 1)
        return local
    # polygon case: the reference re-runs construct_node_map (utils.jl:24-29)
    return _construct_node_map((local != 0).astype(float), np.where(local != 0, polymap, 0))


def _scatter(values, local_nodemap):
    """_create_current_maps (raster branch, out.jl:150-176) / _create_voltage_map (out.jl:418-432)."""
    out = np.zeros(local_nodemap.shape)
    m = local_nodemap > 0
    out[m] = np.asarray(values, dtype=np.float64)[local_nodemap[m] - 1]
    return out


def _process_grid(cmap, cellmap, log_transform, set_null_to_nodata):
    """process_grid! (out.jl:305-319)."""
    if log_transform:
        with np.errstate(divide="ignore", invalid="ignore"):
            cmap = np.where(cmap > 0, np.log10(np.where(cmap > 0, cmap, 1.0)), -9999.0)
    if set_null_to_nodata and cellmap is not None:
        cmap = cmap.copy()
        cmap[np.asarray(cellmap) == 0] = -9999.0
    return cmap


def _node_coords(nodemap, comp):
    """(row, col) of the first cell (column-major order) of every node of `comp` (1-based ids) in `nodemap`."""
    nm = np.asarray(nodemap)
    flat = nm.T.ravel()  # column-major traversal
    cells = np.flatnonzero(flat > 0)
    ids = flat[cells]
    uniq, first = np.unique(ids, return_index=True)
    pos = cells[first]
    rows_all = pos % nm.shape[0]
    cols_all = pos // nm.shape[0]
    idx = np.searchsorted(uniq, comp)
    ok = (idx < len(uniq)) & (uniq[np.minimum(idx, len(uniq) - 1)] == comp)
    if not np.all(ok):
        return None, None
    return rows_all[idx].astype(np.int32), cols_all[idx].astype(np.int32)


def solve(prob, solver, flags, cfg=None, log=True, postprocess=None, stats=None):
    """Pairwise kernel for the HIP backend: core.jl:96-305 restructured the way the batched direct-solver driver is
    (core.jl:367-498): per connected component ONE setup, then the whole pair list goes to the device in one call.

    postprocess(orig_pair, comp, voltages, resistance): optional hook standing where the reference calls
    postprocess() (core.jl:247); voltages are already grounded at the source node (core.jl:231).
    Returns the (P+1) x (P+1) matrix with user ids in row / column 0 (core.jl:299).
    """
    a = prob.G.tocsr()
    T = np.float32 if a.dtype == np.float32 else np.float64
    points = np.asarray(prob.points, dtype=np.int64)
    orig_pts = np.asarray(prob.user_points, dtype=np.int64)
    exclude = set((int(x), int(y)) for x, y in prob.exclude_pairs)
    of = flags.outputflags
    numpoints = len(points)
    resistances = -np.ones((numpoints, numpoints), dtype=T)          # core.jl:130
    voltmatrix = np.zeros((numpoints, numpoints), dtype=T)
    shortcut_res = -np.ones((numpoints, numpoints), dtype=T)
    get_shortcut = (flags.is_raster and not of.write_volt_maps and not of.write_cur_maps and
                    not of.write_cum_cur_map_only and not of.write_max_cur_maps and len(exclude) == 0)  # core.jl:137-146
    want_volt = postprocess is not None and not get_shortcut
    # scope row N1: current / voltage maps (raster). Node currents, their cumulative sum and maximum are computed on
    # the device; only the per-pair maps the flags ask for come back to the host.
    maps = None
    raster_maps = (flags.is_raster and not get_shortcut and prob.nodemap is not None and np.size(prob.nodemap) and
                   (of.write_cur_maps or of.write_cum_cur_map_only or of.write_max_cur_maps or of.write_volt_maps))
    if raster_maps:
        maps = stats.setdefault("maps", {"cur": {}, "volt": {}}) if stats is not None else {"cur": {}, "volt": {}}
        if prob.cum is None:
            prob.cum = initialize_cum_maps(prob.nodemap, of.write_max_cur_maps)
    # network mode: per-pair node / branch current tables and voltages, cumulative vectors (out.jl:46-84)
    network_tables = (not flags.is_raster) and (of.write_cur_maps or of.write_volt_maps)
    if network_tables:
        tables = stats.setdefault("tables", {}) if stats is not None else {}
        ncoords = len(prob.net_coords) if prob.net_coords is not None else 0
        net_cum = stats.setdefault("net_cum", {"branch": np.zeros(ncoords), "node": np.zeros(a.shape[0])}) \
            if stats is not None else {"branch": np.zeros(ncoords), "node": np.zeros(a.shape[0])}
        coord_index = {}
        if prob.net_coords is not None:
            for k, (ci_, cj_) in enumerate(prob.net_coords):
                coord_index.setdefault((int(ci_), int(cj_)), k)
    nsolves = 0
    for comp in prob.cc:
        comp = np.asarray(comp, dtype=np.int64)
        in_comp = np.isin(points, comp)
        csub = []
        for p in points[in_comp]:
            if p not in csub:
                csub.append(int(p))                                   # filter |> unique, core.jl:151
        if not csub:
            continue
        idx0 = comp - 1
        matrix = regularize(a[idx0][:, idx0])                         # core.jl:135,161
        local = {int(node): int(np.searchsorted(comp, node)) for node in csub}
        node_row = node_col = None
        if flags.is_raster and prob.nodemap is not None and np.size(prob.nodemap):
            node_row, node_col = _node_coords(prob.nodemap, comp)
        # ---- pair list: one solve per distinct (src_node, dst_node), core.jl:182-229
        src_nodes, dst_nodes, fan = [], [], []
        last_src = 1 if get_shortcut else len(csub)                   # shortcut: anchor point only, core.jl:256-260
        for a_i in range(last_src):
            src_node = csub[a_i]
            src_indices = np.flatnonzero(points == src_node)
            if not get_shortcut:
                # smash_repeats!, core.jl:188-189,588-603. In shortcut mode the reference discards these entries
                # (the task's result list is ignored, core.jl:259), so ids sharing the ANCHOR's node keep -1 there;
                # that behaviour is reproduced as is.
                for x in range(len(src_indices)):
                    for y in range(x + 1, len(src_indices)):
                        resistances[src_indices[x], src_indices[y]] = 0
                        resistances[src_indices[y], src_indices[x]] = 0
            for b_i in range(a_i + 1, len(csub)):
                dst_node = csub[b_i]
                dst_indices = np.flatnonzero(points == dst_node)
                combos = [(int(ci), int(cj)) for ci in src_indices for cj in dst_indices
                          if (int(orig_pts[ci]), int(orig_pts[cj])) not in exclude]
                if not combos:
                    continue
                src_nodes.append(local[src_node])
                dst_nodes.append(local[dst_node])
                fan.append(combos)
        if src_nodes:
            gather = None
            if get_shortcut:
                focal_in_comp = np.flatnonzero(in_comp)
                gather = np.array([int(np.searchsorted(comp, points[i])) for i in focal_in_comp], dtype=np.int64)
            with construct_cholesky_factor(matrix, solver, node_row, node_col) as factor:   # core.jl:164 (once per CC)
                try:
                    if raster_maps:
                        R, gathered, V, st = _solve_pairs_with_maps(factor, prob, comp, src_nodes, dst_nodes, fan, orig_pts,
                                                                   of, maps, want_volt, bs=getattr(solver, "bs", 16))
                    elif network_tables:
                        # chunks of solver.bs pairs: host memory O((n + nnz) * bs), never (n + nnz) x npairs
                        gathered = None
                        V = np.zeros((matrix.shape[0], len(src_nodes)), dtype=factor.dtype, order="F") if want_volt else None
                        mcsr = matrix.tocsr()
                        mcsr.sort_indices()
                        rows_of = np.repeat(np.arange(mcsr.shape[0]), np.diff(mcsr.indptr))
                        upper = mcsr.indices > rows_of                      # stored entries (row < col), CSR order
                        R = np.zeros(len(src_nodes), dtype=factor.dtype)
                        st = None
                        cbs = max(1, int(getattr(solver, "bs", 16)))
                        for lo in range(0, len(src_nodes), cbs):
                            hi = min(lo + cbs, len(src_nodes))
                            Rc, Vc, C, stc, Bc = factor.solve_pairs_currents(src_nodes[lo:hi], dst_nodes[lo:hi],
                                                                             want_voltages=True, want_currents=True,
                                                                             want_branch=True)
                            R[lo:hi] = Rc
                            if V is not None:
                                V[:, lo:hi] = Vc
                            if st is None:
                                st = dict(stc)
                            else:
                                for key_ in ("total_iters", "solve_ms", "device_ms", "not_converged", "nrhs"):
                                    st[key_] += stc[key_]
                                st["max_iters"] = max(st["max_iters"], stc["max_iters"])
                                st["max_relres"] = max(st["max_relres"], stc["max_relres"])
                            for k, p in enumerate(range(lo, hi)):
                                br = np.column_stack([comp[rows_of[upper]], comp[mcsr.indices[upper]], Bc[upper, k]])
                                node = np.column_stack([comp, C[:, k]])
                                volt = np.column_stack([comp, Vc[:, k]])
                                for (ci, cj) in fan[p]:
                                    for row in br:                               # cumulative branch currents by edge
                                        kk = coord_index.get((int(row[0]), int(row[1])), coord_index.get((int(row[1]), int(row[0]))))
                                        if kk is not None:
                                            net_cum["branch"][kk] += row[2]
                                    net_cum["node"][comp - 1] += C[:, k]
                                    tables[(int(orig_pts[ci]), int(orig_pts[cj]))] = {
                                        "branch": br[~np.isclose(br[:, 2], 0.0, atol=1e-6)],   # write_currents, out.jl:117-124
                                        "node": node, "voltages": volt}
                    else:
                        R, gathered, V, st = factor.solve_pairs(src_nodes, dst_nodes, gather=gather,
                                                                want_voltages=want_volt)
                except lib.CsgpuError as e:
                    if e.code == lib.CSGPU_NOT_CONVERGED:
                        _raise_not_converged(e)                       # core.jl:641
                    raise
                if stats is not None:
                    stats.setdefault("batches", []).append(st)
                    stats.setdefault("levels", []).append(factor.info["levels"])
            nsolves += len(src_nodes)
            for p, combos in enumerate(fan):
                for (ci, cj) in combos:
                    resistances[ci, cj] = R[p]
                    resistances[cj, ci] = R[p]
                    if get_shortcut:
                        # update_voltmatrix!, core.jl:685-703 (i = 2:numpoints in the reference's 1-based loop)
                        for gpos, i in enumerate(focal_in_comp):
                            if i >= 1:
                                voltmatrix[i, cj] = 1 - gathered[p, gpos] / R[p]
                    elif postprocess is not None:
                        postprocess((int(orig_pts[ci]), int(orig_pts[cj])), comp, V[:, p], R[p])
        if get_shortcut:
            anchor = int(np.flatnonzero(points == csub[0])[0])
            _update_shortcut_resistances(anchor, voltmatrix, shortcut_res, resistances, in_comp)
    if get_shortcut:
        resistances = shortcut_res                                    # core.jl:290-292
    np.fill_diagonal(resistances, 0)                                  # core.jl:294-296
    r = np.zeros((numpoints + 1, numpoints + 1), dtype=T)
    r[0, 1:] = orig_pts
    r[1:, 0] = orig_pts
    r[1:, 1:] = resistances
    if stats is not None:
        stats["nsolves"] = stats.get("nsolves", 0) + nsolves
        stats["shortcut"] = bool(get_shortcut)
    if cfg is not None and cfg.get("output_file"):
        save_resistances(r, cfg["output_file"])
    return r


def _solve_pairs_with_maps(factor, prob, comp, src_nodes, dst_nodes, fan, orig_pts, of, maps, want_volt, bs=16):
    """postprocess() with maps on (core.jl:655-683 -> out.jl:29-115): voltage maps, per-pair current maps, cumulative and
    maximum current maps. Node currents come from the device (csgpu_solve_pairs_currents), in CHUNKS of `bs` pairs: the
    cumulative / maximum node currents are accumulated on the device across the chunks, per-pair vectors cross PCIe only
    when per-pair maps are asked for, so host memory is O(n * bs) like the reference's batched driver (core.jl:448-493) --
    unless the caller's postprocess hook wants every voltage vector (want_volt), which then is what it asked for.
    julia/CircuitscapeHIPExt.jl::solve_pairs_with_maps! is the same routine on the reference's side."""
    n = len(comp)
    npairs = len(src_nodes)
    local_nodemap = construct_local_node_map(prob.nodemap, comp, prob.polymap)       # core.jl:170
    per_pair_cur = (of.write_cur_maps and not of.write_cum_cur_map_only) or of.log_transform_maps
    need_volt = of.write_volt_maps or want_volt
    weights = np.array([len(c) for c in fan], dtype=np.int32)   # the reference post-processes once per id combination
    linear = not of.log_transform_maps
    node_cum = np.zeros(n, dtype=factor.dtype) if linear else None     # the handle's value type (float32 problems too)
    node_max = np.zeros(n, dtype=factor.dtype) if (linear and prob.cum.max_curr is not None) else None
    cum = prob.cum
    R = np.zeros(npairs, dtype=factor.dtype)
    Vall = np.zeros((n, npairs), dtype=factor.dtype, order="F") if want_volt else None
    st = None
    bs = max(1, int(bs))
    for lo in range(0, npairs, bs):
        hi = min(lo + bs, npairs)
        Rc, V, C, stc = factor.solve_pairs_currents(src_nodes[lo:hi], dst_nodes[lo:hi], weights=weights[lo:hi],
                                                    want_voltages=need_volt, want_currents=per_pair_cur, cum=node_cum,
                                                    mx=node_max)
        R[lo:hi] = Rc
        if Vall is not None:
            Vall[:, lo:hi] = V
        if st is None:
            st = dict(stc)
        else:
            for k in ("total_iters", "solve_ms", "device_ms", "cg_spmv_ms", "cg_spmv_calls", "not_converged",
                      "graph_launches", "polished_batches", "nrhs"):
                st[k] += stc[k]
            st["max_iters"] = max(st["max_iters"], stc["max_iters"])
            st["max_relres"] = max(st["max_relres"], stc["max_relres"])
        for k, p in enumerate(range(lo, hi)):
            cm = vm = None
            if per_pair_cur:
                cm = _process_grid(_scatter(C[:, k], local_nodemap), prob.cellmap, of.log_transform_maps,
                                   of.set_null_currents_to_nodata)
            if of.write_volt_maps:
                vm = _process_grid(_scatter(V[:, k], local_nodemap), prob.cellmap, False, of.set_null_voltages_to_nodata)
            for (ci, cj) in fan[p]:
                key = (int(orig_pts[ci]), int(orig_pts[cj]))
                if cm is not None:
                    if of.write_cur_maps and not of.write_cum_cur_map_only:
                        maps["cur"][key] = cm
                    if not linear:
                        cum.cum_curr += cm
                        if cum.max_curr is not None:
                            np.maximum(cum.max_curr, cm, out=cum.max_curr)
                if vm is not None:
                    maps["volt"][key] = vm
    if linear:
        cmap = _scatter(node_cum, local_nodemap)
        if of.set_null_currents_to_nodata and prob.cellmap is not None:
            cmap[np.asarray(prob.cellmap) == 0] = -9999.0 * weights.sum()
        cum.cum_curr += cmap
        if cum.max_curr is not None:
            mmap = _process_grid(_scatter(node_max, local_nodemap), prob.cellmap, False, of.set_null_currents_to_nodata)
            np.maximum(cum.max_curr, mmap, out=cum.max_curr)
    return R, None, Vall, st


def _update_shortcut_resistances(anchor, voltmatrix, shortcut, resistances, check):
    """core.jl:706-739 (R_xj = 2 R_aj V_xj + R_ax - R_aj)."""
    l = resistances.shape[0]
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


def single_ground_all_pairs(prob, flags, cfg=None, log=True, **kw):
    """core.jl:70-72."""
    return solve(prob, prob.solver, flags, cfg, log, **kw)


def raster_pairwise_on_device(cellmap, points_rc, solver, four_neighbors=False, avg_res=False, exclude_pairs=(),
                              stats=None, cum=None, polymap=None):
    """Pairwise mode for a raster with the whole graph layer on the device (scope row N4), short-circuit polygons
    included (`polymap`: cells of a polygon share one node, csgpu_raster_setup_poly):
    csgpu_raster_setup numbers the valid cells, writes the CSR Laplacian in HBM and regularises it (core.jl:161),
    csgpu_components labels the connected components, and every solvable pair goes to csgpu_solve_pairs in ONE
    call on ONE handle (the Laplacian of all components is block diagonal; a pair's right-hand side lives in one
    block). No n-sized array is built on the host except the node map it asks for.

    points_rc: (rows, cols, ids), 1-based, unique ids. Returns the padded resistance matrix of
    single_ground_all_pairs (core.jl:130,294-299): -1 for pairs in different components or excluded, 0 on the
    diagonal and for focal points sharing a node. With `cum` (a Cumulative from initialize_cum_maps) the node currents
    of every solved pair are accumulated on the device as well (N1: cumulative / maximum current maps) and scattered
    into cum.cum_curr / cum.max_curr through the device-built node map."""
    rows = np.asarray(points_rc[0], dtype=np.int64) - 1
    cols = np.asarray(points_rc[1], dtype=np.int64) - 1
    ids = np.asarray(points_rc[2], dtype=np.int64)
    with lib.raster_setup(np.asarray(cellmap), _opts_for(solver), four_neighbors=four_neighbors,
                          avg_resistances=avg_res, reg=True, polymap=polymap) as h:
        nodemap = h.raster_nodemap()
        node = nodemap[rows, cols].astype(np.int64)          # 1-based, 0 = focal point on NODATA
        labels, _ = h.components()
        comp = np.where(node > 0, labels[np.maximum(node, 1) - 1], -1)
        npt = len(ids)
        excl = {(int(a), int(b)) for a, b in exclude_pairs} | {(int(b), int(a)) for a, b in exclude_pairs}
        res = -np.ones((npt, npt))
        np.fill_diagonal(res, 0.0)
        pi, pj = [], []
        for i in range(npt):
            for j in range(i + 1, npt):
                if node[i] == 0 or node[j] == 0 or comp[i] != comp[j] or (int(ids[i]), int(ids[j])) in excl:
                    continue
                if node[i] == node[j]:
                    res[i, j] = res[j, i] = 0.0
                    continue
                pi.append(i)
                pj.append(j)
        if pi:
            try:
                if cum is None:
                    R, _, _, st = h.solve_pairs(node[pi] - 1, node[pj] - 1)
                else:
                    n_nodes = h.info["n"]
                    node_cum = np.zeros(n_nodes, dtype=h.dtype)
                    node_max = np.zeros(n_nodes, dtype=h.dtype) if cum.max_curr is not None else None
                    R, _, _, st = h.solve_pairs_currents(node[pi] - 1, node[pj] - 1, want_currents=False, cum=node_cum,
                                                         mx=node_max)
                    cum.cum_curr += _scatter(node_cum, nodemap)
                    if node_max is not None:
                        np.maximum(cum.max_curr, _scatter(node_max, nodemap), out=cum.max_curr)
            except lib.CsgpuError as e:
                if e.code == lib.CSGPU_NOT_CONVERGED:
                    _raise_not_converged(e)
                raise
            if stats is not None:
                stats.update(st)
            res[pi, pj] = R
            res[pj, pi] = R
    out = np.zeros((npt + 1, npt + 1))
    out[0, 1:] = ids
    out[1:, 0] = ids
    out[1:, 1:] = res
    return out


def focal_regions_pairwise_on_device(cellmap, points_rc, solver, four_neighbors=False, avg_res=False, exclude_pairs=(),
                                     stats=None, polymap=None):
    """Pairwise mode when focal points are REGIONS (several cells share an id): the reference short-circuits the two
    regions of every pair and builds a fresh graph + AMG hierarchy per pair (`_pt_file_polygons_path`,
    src/raster/pairwise.jl:72-135 -> create_new_polymap :369-442 -> construct_node_map). Here ONE graph (user polygons
    merged on the device, csgpu_raster_setup[_poly]) and ONE hierarchy serve all pairs: a short-circuited set is an
    equipotential, so the effective resistance between the sets I and J is 1 / (total current leaving I when I is held at
    potential 1 and J at 0) -- a Dirichlet problem on the graph in which they are NOT merged,
        A_ff x_f = -A_fI 1,   x = 1 on I, 0 on J,   R = 1 / sum_{i in I} (A x)_i = 1 / x'Ax,
    which csgpu_solve_region_pairs solves on the device (rows / columns of I u J masked as in csgpu_solve_grounded, one
    column per pair, batches of `solver.bs` pairs per PCG; no n-sized array crosses the boundary). Edges inside a set carry no current at equal potential (the merged graph drops them as
    self-loops); parallel edges from a set to an outside cell add up in the sum (the merged graph sums them). Sets that
    reach several components behave like the merged graph: only components both sets touch carry current.

    Which nodes form a region's set follows create_new_polymap to the letter: a single-cell region is its cell's node; a
    region none of whose cells lies in a user polygon is all of its cells (:401-411); a region with cells in user
    polygons merges THOSE POLYGONS and nothing else (:426-433), the pair being measured from the region's first cell
    (:154-157) -- if that cell lies outside the merged polygons the short-circuit is a floating one, which a Dirichlet
    mask cannot express: such a pair takes the reference's own route (its polygon map built, csgpu_raster_setup_poly and
    csgpu_solve_pairs for that pair alone). Returns the reference's padded resistance matrix over the region ids in order
    of first appearance (-1: not connected or excluded, 0: sets sharing a node)."""
    rows = np.asarray(points_rc[0], dtype=np.int64) - 1
    cols = np.asarray(points_rc[1], dtype=np.int64) - 1
    ids = [int(v) for v in points_rc[2]]
    pts = []
    for v in ids:
        if v not in pts:
            pts.append(v)
    pm = None if polymap is None or np.size(polymap) == 0 else np.asarray(polymap, dtype=np.int64)
    excl = {(int(a), int(b)) for a, b in exclude_pairs} | {(int(b), int(a)) for a, b in exclude_pairs}
    res = -np.ones((len(pts), len(pts)))
    np.fill_diagonal(res, 0.0)
    fallback = []
    try:
        with lib.raster_setup(np.asarray(cellmap), _opts_for(solver), four_neighbors=four_neighbors,
                              avg_resistances=avg_res, reg=True, polymap=pm) as h:
            nodemap = h.raster_nodemap()
            n = h.info["n"]
            labels, _ = h.components()
            cell_node = nodemap[rows, cols].astype(np.int64) - 1        # -1: the cell is NODATA
            region, floating = {}, set()
            for p in pts:
                cells = [k for k in range(len(ids)) if ids[k] == p]
                in_poly = [k for k in cells if pm is not None and pm[rows[k], cols[k]] != 0]
                if len(cells) == 1 or not in_poly:
                    nodes = {int(cell_node[k]) for k in cells if cell_node[k] >= 0}
                else:
                    if len(in_poly) == 1:
                        raise ValueError("focal region %d has exactly one cell inside a polygon: the reference itself "
                                         "fails on this input (undefined variable at src/raster/pairwise.jl:424)" % p)
                    vals = {int(pm[rows[k], cols[k]]) for k in in_poly}
                    sel = np.isin(pm, list(vals)) & (nodemap > 0)
                    nodes = {int(v) - 1 for v in np.unique(nodemap[sel])}
                    if int(cell_node[cells[0]]) not in nodes:
                        floating.add(p)
                region[p] = sorted(nodes)
            jobs = []
            for a in range(len(pts)):
                for b in range(a + 1, len(pts)):
                    if (pts[a], pts[b]) in excl:
                        continue
                    if pts[a] in floating or pts[b] in floating:
                        fallback.append((a, b))
                        continue
                    I, J = region[pts[a]], region[pts[b]]
                    common = {int(labels[v]) for v in I} & {int(labels[v]) for v in J}
                    if not common:
                        continue                                       # stays -1
                    if set(I) & set(J):
                        res[a, b] = res[b, a] = 0.0
                        continue
                    jobs.append((a, b, [v for v in I if int(labels[v]) in common],
                                 [v for v in J if int(labels[v]) in common]))
            if jobs:
                # the sets of every job go down once; indicator, right-hand side, masked solve and energy stay on the device
                sets, src_set, dst_set = [], [], []
                for (_, _, I, J) in jobs:
                    src_set.append(len(sets))
                    sets.append(I)
                    dst_set.append(len(sets))
                    sets.append(J)
                R, st = h.solve_region_pairs(sets, src_set, dst_set)
                if stats is not None:
                    stats.update({k: st[k] for k in ("total_iters", "nrhs", "max_relres")})
                for (a, b, _, _), r in zip(jobs, R):
                    res[a, b] = res[b, a] = float(r)
    except lib.CsgpuError as e:
        if e.code == lib.CSGPU_NOT_CONVERGED:
            _raise_not_converged(e)
        raise
    if fallback:
        if stats is not None:
            stats["per_pair_graphs"] = len(fallback)
        for a, b in fallback:
            # the pair's polygon map as create_new_polymap builds it (pt1 / pt2 branch, pairwise.jl:406-440): a region
            # clear of user polygons becomes a new polygon, a region with cells in user polygons merges those polygons
            newpoly = pm.copy()
            nxt = int(pm.max())
            for p in (pts[a], pts[b]):
                cells = [k for k in range(len(ids)) if ids[k] == p]
                if len(cells) == 1:
                    continue
                in_poly = [k for k in cells if pm[rows[k], cols[k]] != 0]
                nxt += 1
                if not in_poly:
                    for k in cells:
                        newpoly[rows[k], cols[k]] = nxt
                else:
                    newpoly[np.isin(pm, [int(pm[rows[k], cols[k]]) for k in in_poly])] = nxt
            first = [ids.index(pts[a]), ids.index(pts[b])]
            pair_pts = ([int(rows[k]) + 1 for k in first], [int(cols[k]) + 1 for k in first], [pts[a], pts[b]])
            r2 = raster_pairwise_on_device(cellmap, pair_pts, solver, four_neighbors=four_neighbors, avg_res=avg_res,
                                           polymap=newpoly)
            res[a, b] = res[b, a] = r2[1, 2]
    out = np.zeros((len(pts) + 1, len(pts) + 1))
    out[0, 1:] = pts
    out[1:, 0] = pts
    out[1:, 1:] = res
    return out


def compute_3col(r):
    """out.jl:12-26."""
    fp = r[1:, 0]
    l = len(fp)
    out = np.zeros((l * (l - 1) // 2, 3), dtype=r.dtype)
    k = 0
    for i in range(l):
        for j in range(i + 1, l):
            out[k] = (fp[i], fp[j], r[j + 1, i + 1])
            k += 1
    return out


def save_resistances(r, output_file):
    """out.jl:454-465: <prefix>_resistances.out and <prefix>_resistances_3columns.out."""
    pref = output_file.split(".out")[0]
    np.savetxt(pref + "_resistances.out", r, delimiter=" ", fmt="%.10g")
    np.savetxt(pref + "_resistances_3columns.out", compute_3col(r), delimiter=" ", fmt="%.10g")
