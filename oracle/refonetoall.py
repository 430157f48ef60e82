"""TEST INFRASTRUCTURE ONLY -- restatement of the reference's one-to-all / all-to-one raster drivers (scope row N2).

Restates (Circuitscape.jl, paths relative to /root/reference):
  onetoall_kernel, prune_points!, prune_strengths     src/raster/onetoall.jl:13-194
  create_new_polymap (point_map branch)               src/raster/pairwise.jl:369-404
  advanced_kernel (raster branch, check_node, one-to-all / all-to-one return values, per-point maps)
                                                      src/raster/advanced.jl:151-271
  write_cum_maps / postprocess                        src/out.jl:467-481
The restatement is literal, including the reference's behaviour with an included-pairs file (the node map is
rebuilt from the ORIGINAL polygon map while the Laplacian of the combined polygon map is kept, onetoall.jl:91-93)
and its use of the loop index as an index into the point list for `check_node` (onetoall.jl:124).
"""
import numpy as np
import scipy.sparse as sp

from . import refgraph as rg
from . import refmaps as rm
from . import refsolve as rs


def create_new_polymap_pointmap(polymap, points_rc, point_map):
    """pairwise.jl:374-404: combine the polygon map and the focal-point map."""
    if polymap is None or np.size(polymap) == 0:
        return point_map
    newpoly = np.array(polymap, dtype=np.int64, copy=True)
    ids = list(points_rc[2])
    ii, jj = rg._colmajor_nonzero(point_map != 0)
    if len(ids) == len(set(ids)):
        k = int(polymap.max())
        for i, j in zip(ii, jj):
            if polymap[i, j] == 0:
                newpoly[i, j] = point_map[i, j] + k
    else:
        k = max(int(polymap.max()), int(point_map.max()))
        for i, j in zip(ii, jj):
            v1, v2 = point_map[i, j], newpoly[i, j]
            if v2 == 0:
                newpoly[i, j] = k + v1
                continue
            if v1 != v2:
                newpoly[newpoly == v2] = v1
    return newpoly


def _advanced_kernel_raster(G, cc, nodemap, polymap, sources, grounds, finitegrounds, check_node, source_map, gmap,
                            is_onetoall, solve, opts):
    """advanced.jl:151-271, raster branch. Returns (ret, outcurr, voltmap, curmap)."""
    outvolt = np.zeros(gmap.shape)
    outcurr = np.zeros(gmap.shape)
    volt = np.zeros(gmap.shape)
    voltages = np.zeros(G.shape[0])
    no_finite = len(finitegrounds) == 1 and finitegrounds[0] == -9999
    solver_called = False
    for c in cc:
        if check_node != -1 and check_node not in c:
            continue
        idx = np.asarray(c) - 1
        s_local, g_local = sources[idx], grounds[idx]
        if s_local.sum() == 0 or g_local.sum() == 0:
            continue
        f_local = finitegrounds if no_finite else finitegrounds[idx]
        a_local = G[idx][:, idx]
        voltages[idx] += rs.multiple_solver(a_local, s_local, g_local, f_local, solve)
        local_nodemap = rm.construct_local_node_map(nodemap, c, polymap)
        solver_called = True
        outvolt += rm.scatter(voltages[idx], local_nodemap)
        outcurr += rm.scatter(rm.get_node_currents_grounded(a_local, voltages[idx], f_local), local_nodemap)
        m = local_nodemap > 0
        volt[m] = voltages[idx][local_nodemap[m] - 1]
    voltmap = rm.process_grid(outvolt, gmap, False, opts["set_null_voltages_to_nodata"])
    curmap = rm.process_grid(outcurr, gmap, opts["log_transform_maps"], opts["set_null_currents_to_nodata"])
    if not solver_called:
        return -1.0, outcurr, voltmap, curmap
    if is_onetoall:
        ii, jj = rg._colmajor_nonzero(source_map != 0)
        val = volt[ii, jj] / source_map[ii, jj]
        return (-1.0 if val[0] == 0 else float(val[0])), outcurr, voltmap, curmap
    return 0.0, outcurr, voltmap, curmap


def onetoall_from_fixture(case, mode="direct", solve=None):
    """raster_one_to_all (onetoall.jl:1-11) on a tests/golden oneToAllVerify / allToOneVerify fixture.
    Returns {'res': [[id, value]...], 'cum', 'max' (or None), 'points': {id: {'voltmap', 'curmap'}}}."""
    o = case["options"]
    one_to_all = case["kind"] == "one_to_all"
    gmap = np.asarray(case["cellmap"], dtype=np.float64)
    polymap = np.asarray(case["polymap"], dtype=np.int64) if case.get("polymap") is not None else None
    pr = [list(x) for x in case["points_rc"]]
    strengths = np.array(case["strengths"], dtype=np.float64) if case.get("strengths") else None
    inc = case.get("included_pairs")
    use_var = strengths is not None
    use_inc = inc is not None
    mode_flag = 0 if (use_inc and inc["mode"] == "include") else 1
    if use_inc:
        ids = list(inc["point_ids"])
        keep = [k for k, p in enumerate(pr[2]) if p in ids]          # prune_points!
        pr = [[col[k] for k in keep] for col in pr]
        if use_var:
            strengths = strengths[[k for k, p in enumerate(strengths[:, 0]) if p in ids]]  # prune_strengths
    npts = len(pr[0])
    point_map = np.zeros(gmap.shape, dtype=np.int64)
    for x in range(npts):
        point_map[pr[0][x] - 1, pr[1][x] - 1] = pr[2][x]
    points_unique = list(dict.fromkeys(pr[2]))
    newpoly = create_new_polymap_pointmap(polymap, pr, point_map)
    nodemap = rg.construct_node_map(gmap, newpoly)
    a = rg.construct_graph(gmap, nodemap, o["connect_using_avg_resistances"], o["connect_four_neighbors_only"])
    cc = rg.connected_components(a)
    G = sp.csr_matrix(rg.laplacian(a))
    unique_point_map = np.zeros(gmap.shape, dtype=np.int64)
    for i in points_unique:
        ind = pr[2].index(i)
        unique_point_map[pr[0][ind] - 1, pr[1][ind] - 1] = pr[2][ind]
    solve = solve or rs._oracle_multiple_solve(mode)
    res = np.zeros(len(points_unique))
    cum = np.zeros(gmap.shape)
    mx = np.full(gmap.shape, -9999.0) if o["write_max_cur_maps"] else None
    out_points = {}
    original_point_map = point_map
    for i, n in enumerate(points_unique):
        point_map = original_point_map.copy()
        nodemap_i, newpoly_i = nodemap, newpoly
        strn = strengths[i, 1] if use_var else 1.0
        if use_inc:
            mat = np.asarray(inc["matrix"])
            for j in range(len(ids)):
                if i != j and mat[i, j] == mode_flag:
                    point_map[point_map == ids[j]] = 0
            newpoly_i = create_new_polymap_pointmap(polymap, pr, point_map)
            nodemap_i = rg.construct_node_map(gmap, polymap)   # sic: the original polygon map (onetoall.jl:92)
        strength_map = None
        if use_var:
            tmp = np.array([point_map[pr[0][x] - 1, pr[1][x] - 1] for x in range(npts)])
            st = strengths.copy()
            st[np.flatnonzero(tmp == 0), 1] = 1
            strength_map = np.zeros(gmap.shape)
            for x in range(npts):
                strength_map[pr[0][x] - 1, pr[1][x] - 1] = st[x, 1]
        if point_map.sum() == n:
            res[i] = -1
            continue
        if one_to_all:
            source_map = np.where(unique_point_map == n, float(strn), 0.0)
            ground_map = np.where(point_map == n, 0.0, point_map.astype(np.float64))
            ground_map[ground_map > 0] = np.inf
        else:
            if use_var:
                source_map = np.where(unique_point_map == n, 0.0, strength_map)
            else:
                source_map = np.where(unique_point_map != 0, 1.0, 0.0)
                source_map = np.where(point_map == n, 0.0, source_map)
            ground_map = np.where(point_map == n, np.inf, 0.0)
        check_node = nodemap_i[pr[0][i] - 1, pr[1][i] - 1]
        policy = "rmvgnd" if one_to_all else "rmvsrc"
        nn = G.shape[0]
        sources = np.zeros(nn)
        grounds = np.zeros(nn)
        for smap, acc in ((source_map, sources), (ground_map, grounds)):
            ii, jj = rg._colmajor_nonzero(smap != 0)
            for r, c in zip(ii, jj):
                v = nodemap_i[r, c]
                if v != 0:
                    acc[v - 1] += smap[r, c]
        sources, grounds, finite = rs.resolve_conflicts(sources, grounds, policy)
        ret, curr, voltmap, curmap = _advanced_kernel_raster(G, cc, nodemap_i, newpoly_i, sources, grounds, finite,
                                                             check_node, source_map, gmap, one_to_all, solve, o)
        res[i] = ret
        out_points[str(n)] = {"voltmap": voltmap, "curmap": curmap}
        cum += curr
        if mx is not None:
            mx = np.maximum(mx, curr)
    cum = rm.process_grid(cum, gmap, o["log_transform_maps"], o["set_null_currents_to_nodata"])
    if mx is not None:
        mx = rm.process_grid(mx, gmap, o["log_transform_maps"], o["set_null_currents_to_nodata"])
    return {"res": np.column_stack([np.array(points_unique, dtype=np.float64), res]), "cum": cum, "max": mx,
            "points": out_points}
