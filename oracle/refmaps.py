"""TEST INFRASTRUCTURE ONLY -- restatement of the reference's per-pair map post-processing (scope row N1).

Restates (Circuitscape.jl, paths relative to /root/reference):
  _get_branch_currents_posneg / _get_branch_currents   src/out.jl:209-290
  _get_node_currents_posneg / get_node_currents        src/out.jl:178-207
  _create_current_maps (raster branch)                 src/out.jl:150-176
  _create_voltage_map                                  src/out.jl:418-432
  construct_local_node_map                             src/utils.jl:10-30
  write_cur_maps accumulation (cum / max)              src/out.jl:86-115, src/utils.jl:114-142
  raster_pairwise drivers with maps on                 src/raster/pairwise.jl:55-135, src/core.jl:655-683
"""
import numpy as np
import scipy.sparse as sp

from . import refgraph as rg
from . import refsolve as rs


def get_node_currents(G, voltages):
    """out.jl:178-207 for finitegrounds == [-9999] (pairwise mode)."""
    G = sp.csr_matrix(G)
    coo = sp.triu(G, k=1).tocoo()  # entries (row < col): the reference's `i > row` orientation
    g = np.abs(coo.data)
    v = np.asarray(voltages, dtype=np.float64)
    n = G.shape[0]
    out = []
    for pos in (True, False):
        b = g * (v[coo.row] - v[coo.col]) if pos else g * (v[coo.col] - v[coo.row])
        maxcur = b.max() if len(b) else 1.0
        with np.errstate(divide="ignore", invalid="ignore"):
            b = np.where(np.abs(b / maxcur) < 1e-8, 0.0, b)
        B = sp.coo_matrix((b, (coo.row, coo.col)), shape=(n, n)).tocsr()
        C = (B - B.T).tocsr()
        C.data[C.data < 0] = 0.0
        out.append(np.asarray(C.sum(axis=0)).ravel())
    return np.maximum(out[0], out[1])


def construct_local_node_map(nodemap, component, polymap):
    """utils.jl:10-30 (component: 1-based node ids)."""
    nodemap = np.asarray(nodemap)
    local = np.where(np.isin(nodemap, component), nodemap, 0)
    if np.array_equal(local, nodemap):
        return local
    if polymap is None or np.size(polymap) == 0:
        ii, jj = rg._colmajor_nonzero(local != 0)
        local = local.copy()
        local[ii, jj] = np.arange(1, len(ii) + 1)
        return local
    local_poly = np.where(local != 0, polymap, 0)
    return rg.construct_node_map((local != 0).astype(float), local_poly)


def scatter(values, local_nodemap):
    """_create_current_maps raster branch / _create_voltage_map: map[i,j] = values[nodemap[i,j]] (0 where no node)."""
    out = np.zeros(local_nodemap.shape)
    m = local_nodemap > 0
    out[m] = np.asarray(values)[local_nodemap[m] - 1]
    return out


def process_grid(cmap, cellmap, log_transform, set_null_to_nodata):
    """process_grid! / write_grid options, out.jl:305-319,355-365."""
    cmap = np.array(cmap, dtype=np.float64)
    if log_transform:
        with np.errstate(divide="ignore", invalid="ignore"):
            cmap = np.where(cmap > 0, np.log10(np.where(cmap > 0, cmap, 1.0)), -9999.0)
    if set_null_to_nodata:
        cmap[np.asarray(cellmap) == 0] = -9999.0
    return cmap


def raster_pairwise_maps_from_fixture(case, mode="direct"):
    """Runs the pairwise driver with maps on (as the fixture's INI asks) and returns
    {'cum': cum_curmap, 'max': max_curmap or None, 'cur': {(a,b): map}, 'volt': {(a,b): map}, 'R': padded matrix}."""
    o = case["options"]
    gmap = np.array(case["cellmap"], dtype=np.float64)
    polymap = np.array(case["polymap"], dtype=np.int64) if case["polymap"] is not None else None
    points_rc = tuple(list(x) for x in case["points_rc"])
    avg_res, four = o["connect_using_avg_resistances"], o["connect_four_neighbors_only"]
    cum = np.zeros(gmap.shape)
    mx = np.full(gmap.shape, -9999.0) if o["write_max_cur_maps"] else None
    cur, volt = {}, {}

    def process(prob):
        a = prob.G.tocsr()
        points = [int(p) for p in prob.points]
        orig = [int(p) for p in prob.user_points]
        exclude = set((int(x), int(y)) for x, y in prob.exclude_pairs)
        for comp in prob.cc:
            compset = {int(x): k for k, x in enumerate(comp)}
            csub = []
            for p in points:
                if p in compset and p not in csub:
                    csub.append(p)
            if not csub:
                continue
            idx0 = np.asarray(comp, dtype=np.int64) - 1
            matrix = rs.regularize(a[idx0][:, idx0])
            solver = rs.DirectSolver(matrix) if mode == "direct" else rs.OracleAMG(matrix)
            local_nodemap = construct_local_node_map(prob.nodemap, np.asarray(comp), prob.polymap)
            for ai in range(len(csub)):
                src_idx = [k for k, p in enumerate(points) if p == csub[ai]]
                for bi in range(ai + 1, len(csub)):
                    dst_idx = [k for k, p in enumerate(points) if p == csub[bi]]
                    combos = [(ci, cj) for ci in src_idx for cj in dst_idx if (orig[ci], orig[cj]) not in exclude]
                    if not combos:
                        continue
                    b = np.zeros(matrix.shape[0])
                    b[compset[csub[ai]]] = -1.0
                    b[compset[csub[bi]]] = 1.0
                    v = rs.solve_linear_system(solver, matrix, b, mode)
                    v = v - v[compset[csub[ai]]]
                    nc = get_node_currents(matrix, v)
                    cmap = process_grid(scatter(nc, local_nodemap), gmap, o.get("log_transform_maps", False),
                                        o.get("set_null_currents_to_nodata", False))
                    vmap = process_grid(scatter(v, local_nodemap), gmap, False, o.get("set_null_voltages_to_nodata", False))
                    for (ci, cj) in combos:
                        nonlocal_cum_add(cmap)
                        cur[(orig[ci], orig[cj])] = cmap
                        volt[(orig[ci], orig[cj])] = vmap

    def nonlocal_cum_add(cmap):
        cum[...] = cum + cmap
        if mx is not None:
            mx[...] = np.maximum(mx, cmap)

    if len(points_rc[0]) == len(set(points_rc[2])):
        prob = rg.compute_graph_data_no_polygons(gmap, polymap, points_rc, case["included_pairs"], avg_res, four)
        process(prob)
    else:
        exclude = set()
        if case["included_pairs"] is not None:
            ex, points_rc = rg.generate_exclude_pairs(points_rc, case["included_pairs"])
            exclude = set(ex)
        pts = []
        for v in points_rc[2]:
            if v not in pts:
                pts.append(v)
        for i in range(len(pts)):
            for j in range(i + 1, len(pts)):
                if (pts[i], pts[j]) in exclude or (pts[j], pts[i]) in exclude:
                    continue
                process(rg.compute_graph_data_polygons(gmap, polymap, points_rc, pts[i], pts[j], avg_res, four))
    cum[cum < -9999] = -9999  # postprocess_cum_curmap!, utils.jl:114-120
    if mx is not None:
        mx[mx < -9999] = -9999
    return {"cum": cum, "max": mx, "cur": cur, "volt": volt}


# ---------------------------------------------------------------------------------------------------------------
# network pairwise with current output: restates write_cur_maps (network branch, out.jl:46-84), _convert_to_3col
# (out.jl:128-148), write_currents (out.jl:117-124), write_voltages (out.jl:410-416), the cumulative vectors
# (utils.jl:133-142) and network_pairwise's cum output (network/pairwise.jl:18-27).
def branch_currents_3col(G, voltages, cc):
    """|B| with B the pos-orientation branch currents (upper triangle, 1e-8*max threshold); rows (cc[row], cc[col], val)."""
    G = sp.csr_matrix(G)
    coo = sp.triu(G, k=1).tocoo()
    v = np.asarray(voltages, dtype=np.float64)
    b = np.abs(coo.data) * (v[coo.row] - v[coo.col])
    maxcur = b.max() if len(b) else 1.0
    with np.errstate(divide="ignore", invalid="ignore"):
        b = np.where(np.abs(b / maxcur) < 1e-8, 0.0, b)
    cc = np.asarray(cc)
    return np.column_stack([cc[coo.row], cc[coo.col], np.abs(b)])


def network_pairwise_tables_from_fixture(case, mode="direct"):
    """Returns {'pairs': {(a,b): {'branch','node','voltages'}}, 'branch_cum', 'node_cum'} with 1-based node ids."""
    prob = rg.compute_graph_data_network(case["edges_i"], case["edges_j"], case["edges_v"], case["focal"])
    a = prob.G.tocsr()
    m = a.shape[0]
    coords = list(zip(case["edges_i"], case["edges_j"]))
    cum_branch = np.zeros(len(coords))
    cum_node = np.zeros(m)
    points = [int(p) for p in prob.points]
    out = {}
    for comp in prob.cc:
        compset = {int(x): k for k, x in enumerate(comp)}
        csub = []
        for p in points:
            if p in compset and p not in csub:
                csub.append(p)
        if not csub:
            continue
        idx0 = np.asarray(comp, dtype=np.int64) - 1
        matrix = rs.regularize(a[idx0][:, idx0])
        solver = rs.DirectSolver(matrix) if mode == "direct" else rs.OracleAMG(matrix)
        for ai in range(len(csub)):
            for bi in range(ai + 1, len(csub)):
                b = np.zeros(matrix.shape[0])
                b[compset[csub[ai]]] = -1.0
                b[compset[csub[bi]]] = 1.0
                v = rs.solve_linear_system(solver, matrix, b, mode)
                v = v - v[compset[csub[ai]]]
                node = get_node_currents(matrix, v)
                br = branch_currents_3col(matrix, v, comp)
                for row in br:
                    key = (int(row[0]), int(row[1]))
                    k = coords.index(key) if key in coords else coords.index((key[1], key[0]))
                    cum_branch[k] += row[2]
                cum_node[idx0] += node
                out[(csub[ai], csub[bi])] = {"branch": br[~np.isclose(br[:, 2], 0.0, atol=1e-6)],
                                              "node": np.column_stack([np.asarray(comp), node]),
                                              "voltages": np.column_stack([np.asarray(comp), v])}
    bc = np.column_stack([np.array(case["edges_i"]), np.array(case["edges_j"]), cum_branch])
    return {"pairs": out, "branch_cum": bc[~np.isclose(bc[:, 2], 0.0, atol=1e-6)],
            "node_cum": np.column_stack([np.arange(1, m + 1), cum_node])}


# ---------------------------------------------------------------------------------------------------------------
# advanced mode, raster flavour (scope row N2). Restates
#   compute_advanced_data / _get_sources_and_grounds (raster branch)   src/raster/advanced.jl:36-116
#   advanced_kernel (raster branch: voltage map, current map)           src/raster/advanced.jl:151-271
#   get_node_currents with finite grounds                               src/out.jl:178-207
def get_node_currents_grounded(G, voltages, finitegrounds):
    """out.jl:178-207 including the finite-ground branch: the current a node sends to ground through its finite
    ground conductance (g_i * v_i) is added to the node's outgoing (v > 0: 'neg' pass) or incoming flow."""
    G = sp.csr_matrix(G)
    fg = np.asarray(finitegrounds, dtype=np.float64)
    if len(fg) == 1 and fg[0] == -9999:
        return get_node_currents(G, voltages)
    coo = sp.triu(G, k=1).tocoo()
    g = np.abs(coo.data)
    v = np.asarray(voltages, dtype=np.float64)
    n = G.shape[0]
    out = []
    for pos in (True, False):
        b = g * (v[coo.row] - v[coo.col]) if pos else g * (v[coo.col] - v[coo.row])
        maxcur = b.max() if len(b) else 1.0
        with np.errstate(divide="ignore", invalid="ignore"):
            b = np.where(np.abs(b / maxcur) < 1e-8, 0.0, b)
        B = sp.coo_matrix((b, (coo.row, coo.col)), shape=(n, n)).tocsr()
        C = (B - B.T).tocsr()
        C.data[C.data < 0] = 0.0
        fc = fg * v
        fc = np.where(fc < 0, -fc, 0.0) if pos else np.where(fc > 0, fc, 0.0)
        out.append(np.asarray(C.sum(axis=0)).ravel() + fc)
    return np.maximum(out[0], out[1])


def _maps_from_fixture(case):
    def arr(m):
        return np.array([[float(x) for x in row] for row in m], dtype=np.float64)
    return arr(case["source_map"]), arr(case["ground_map"])


def raster_advanced_from_fixture(case, mode="direct", solve=None):
    """raster_advanced (raster/advanced.jl:17-34) on a tests/golden mgVerify fixture. Pinned on the reference's mgVerify1..6
    voltage / current map goldens and on mgVerify7_curmap.asc (355 x 481 cells, 5574 finite grounds: the largest golden the
    reference ships; its own suite stops at 6) with the reference's criterion sum(abs2, x - r) < 1e-6 (tests/test_oracle_golden.py).
    Returns {'voltmap': processed voltage map, 'curmap': processed current map, 'volt': raw per-cell voltages}."""
    o = case["options"]
    gmap = np.asarray(case["cellmap"], dtype=np.float64)
    polymap = np.asarray(case["polymap"], dtype=np.int64) if case.get("polymap") is not None else None
    source_map, ground_map = _maps_from_fixture(case)
    nodemap = rg.construct_node_map(gmap, polymap)
    A = rg.construct_graph(gmap, nodemap, o["connect_using_avg_resistances"], o["connect_four_neighbors_only"])
    G = rg.laplacian(A)
    cc = rg.connected_components(A)
    n = G.shape[0]
    sources = np.zeros(n)
    grounds = np.zeros(n)
    for smap, acc in ((source_map, sources), (ground_map, grounds)):
        ii, jj = rg._colmajor_nonzero(smap != 0)
        for i, j in zip(ii, jj):
            v = nodemap[i, j]
            if v != 0:
                acc[v - 1] += smap[i, j]
    sources, grounds, finitegrounds = rs.resolve_conflicts(sources, grounds, o["remove_src_or_gnd"])
    no_finite = len(finitegrounds) == 1 and finitegrounds[0] == -9999
    solve = solve or rs._oracle_multiple_solve(mode)
    outvolt = np.zeros(gmap.shape)
    outcurr = np.zeros(gmap.shape)
    volt = np.zeros(gmap.shape)
    voltages = np.zeros(n)
    G = sp.csr_matrix(G)
    for c in cc:
        idx = np.asarray(c) - 1
        s_local, g_local = sources[idx], grounds[idx]
        if s_local.sum() == 0 or g_local.sum() == 0:
            continue
        f_local = finitegrounds if no_finite else finitegrounds[idx]
        a_local = G[idx][:, idx]
        voltages[idx] += rs.multiple_solver(a_local, s_local, g_local, f_local, solve)
        local_nodemap = construct_local_node_map(nodemap, c, polymap)
        outvolt += scatter(voltages[idx], local_nodemap)
        outcurr += scatter(get_node_currents_grounded(a_local, voltages[idx], f_local), local_nodemap)
        m = local_nodemap > 0
        volt[m] = voltages[idx][local_nodemap[m] - 1]
    return {
        "voltmap": process_grid(outvolt, gmap, False, o["set_null_voltages_to_nodata"]),
        "curmap": process_grid(outcurr, gmap, o["log_transform_maps"], o["set_null_currents_to_nodata"]),
        "volt": volt,
    }


def compute_omniscape_current(conductance, source, ground, four_neighbors=False, mode="direct", solve=None,
                              avg_resistances=False, policy="rmvsrc", want="curmap"):
    """compute_omniscape_current (src/utils.jl:145-257): advanced mode on in-memory rasters -- no polygons, policy
    :rmvsrc, avg_res = false (utils.jl:193-196) -- returning the raw accumulated current map.
    This IS raster_advanced_from_fixture with those options fixed, i.e. the code the reference's mgVerify goldens pin
    (tests/test_oracle_golden.py); `avg_resistances` / `policy` / `want` exist so that the pinning test can drive this
    very function with a fixture's own options (mgVerify2, mgVerify6: the reference cases without polygons)."""
    case = {"options": {"connect_using_avg_resistances": avg_resistances, "connect_four_neighbors_only": four_neighbors,
                        "remove_src_or_gnd": policy, "set_null_voltages_to_nodata": False,
                        "set_null_currents_to_nodata": False, "log_transform_maps": False},
            "cellmap": np.asarray(conductance, dtype=np.float64), "polymap": None,
            "source_map": np.asarray(source, dtype=np.float64), "ground_map": np.asarray(ground, dtype=np.float64)}
    return raster_advanced_from_fixture(case, mode=mode, solve=solve)[want]


def omniscape_moving_window(conductance, source_strength, radius, block_size=1, four_neighbors=False, mode="direct"):
    """CHECKER for solver.omniscape_moving_window: the same windows (restated here, so the product's window generator is
    checked too), every window through compute_omniscape_current above with its centre tied directly to ground
    (ground = inf: multiple_solver deletes the row, raster/advanced.jl:282-288), currents added into the mosaic."""
    cond = np.asarray(conductance, dtype=np.float64)
    strength = np.asarray(source_strength, dtype=np.float64)
    R, C = cond.shape
    half = block_size // 2
    cum = np.zeros(cond.shape)
    nwin = 0
    for ci in range(half, R, block_size):
        for cj in range(half, C, block_size):
            blk = (slice(max(ci - half, 0), min(ci + half + 1, R)), slice(max(cj - half, 0), min(cj + half + 1, C)))
            weight = float(np.where(cond[blk] > 0, strength[blk], 0.0).sum())
            if weight <= 0 or cond[ci, cj] <= 0:
                continue
            win = (slice(max(ci - radius, 0), min(ci + radius + 1, R)), slice(max(cj - radius, 0), min(cj + radius + 1, C)))
            ii, jj = np.mgrid[win]
            disc = (ii - ci) ** 2 + (jj - cj) ** 2 <= radius ** 2
            wc = np.where(disc, cond[win], 0.0)
            full_src = np.where(cond > 0, strength, 0.0)
            full_src[blk] = 0.0
            ws = np.where(disc, full_src[win], 0.0)
            if ws.sum() <= 0:
                continue
            ws = ws * (weight / ws.sum())
            wg = np.zeros(wc.shape)
            wg[ci - win[0].start, cj - win[1].start] = np.inf
            cum[win] += compute_omniscape_current(wc, ws, wg, four_neighbors=four_neighbors, mode=mode)
            nwin += 1
    return cum, nwin
