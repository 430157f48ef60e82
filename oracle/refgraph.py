"""TEST INFRASTRUCTURE ONLY -- restatement of the reference's graph construction (input side of the
hot-path boundary), used to turn the golden fixtures into the `GraphProblem` the solver layer consumes.

Restates (Circuitscape.jl, paths relative to /root/reference):
  construct_node_map      src/raster/pairwise.jl:271-301   (+ relabel! :303-314)
  construct_graph         src/raster/pairwise.jl:316-362   (+ averaging rules :364-367)
  laplacian!              src/core.jl:608-634
  connected_components    Graphs.jl semantics: components ordered by smallest vertex, ascending inside
  create_new_polymap      src/raster/pairwise.jl:369-442   (pt1/pt2 branch only)
  generate_exclude_pairs  src/raster/pairwise.jl:240-269   (+ prune_points! raster/onetoall.jl:169-180)
  compute_graph_data_*    src/raster/pairwise.jl:137-238, src/network/pairwise.jl:31-65

Conventions: node ids, focal ids and `cc` entries are 1-based exactly as in the reference (0 = no node);
matrices are scipy CSR with 0-based storage (row/col = node id - 1).
Nothing in the product imports this module.
"""
from collections import namedtuple
import math

import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import connected_components as _scc

RefProblem = namedtuple(
    "RefProblem", "G cc points user_points exclude_pairs nodemap polymap is_raster"
)

SQRT2 = math.sqrt(2.0)


def res_avg(x, y):
    with np.errstate(divide="ignore"):
        return 1.0 / ((1.0 / x + 1.0 / y) / 2.0) if (x != 0 and y != 0) else 0.0


def cond_avg(x, y):
    return (x + y) / 2.0


def weird_avg(x, y):
    return (x + y) / (2.0 * SQRT2)


def weirder_avg(x, y):
    return 1.0 / (SQRT2 * (1.0 / x + 1.0 / y) / 2.0) if (x != 0 and y != 0) else 0.0


def _colmajor_nonzero(mask):
    """(rows, cols) of True entries in Julia's column-major findall order."""
    jj, ii = np.nonzero(mask.T)
    return ii, jj


def relabel(nodemap, offset=1):
    """pairwise.jl:303-314 -- dense re-ranking of the non-zero labels, order preserving."""
    ii, jj = _colmajor_nonzero(nodemap != 0)
    old = nodemap[ii, jj]
    uniq, inv = np.unique(old, return_inverse=True)
    nodemap[ii, jj] = inv + offset


def construct_node_map(gmap, polymap):
    """pairwise.jl:271-301."""
    gmap = np.asarray(gmap, dtype=np.float64)
    nodemap = np.zeros(gmap.shape, dtype=np.int64)
    ii, jj = _colmajor_nonzero(gmap > 0)
    nodemap[ii, jj] = np.arange(1, len(ii) + 1)
    if polymap is None or polymap.size == 0:
        return nodemap
    polymap = np.asarray(polymap, dtype=np.int64)
    pruned = np.where(gmap > 0, polymap, 0)
    for polynum in np.unique(polymap):
        if polynum == 0:
            continue
        i1, j1 = _colmajor_nonzero(pruned == polynum)
        if len(i1) > 0:
            nodemap[polymap == polynum] = nodemap[i1[0], j1[0]]
    relabel(nodemap, 1)
    return nodemap


def construct_graph(gmap, nodemap, avg_res, four_neighbors):
    """pairwise.jl:316-362 -> symmetric adjacency (scipy CSR, duplicates summed, a + a')."""
    f1 = res_avg if avg_res else cond_avg
    f2 = weirder_avg if avg_res else weird_avg
    nr, nc = gmap.shape
    I, J, V = [], [], []
    for j in range(nc):
        for i in range(nr):
            if nodemap[i, j] == 0:
                continue
            if j != nc - 1 and nodemap[i, j + 1] != 0:
                I.append(nodemap[i, j]); J.append(nodemap[i, j + 1]); V.append(f1(gmap[i, j], gmap[i, j + 1]))
            if i != nr - 1 and nodemap[i + 1, j] != 0:
                I.append(nodemap[i, j]); J.append(nodemap[i + 1, j]); V.append(f1(gmap[i, j], gmap[i + 1, j]))
            if not four_neighbors:
                if i != nr - 1 and j != nc - 1 and nodemap[i + 1, j + 1] != 0:
                    I.append(nodemap[i, j]); J.append(nodemap[i + 1, j + 1]); V.append(f2(gmap[i, j], gmap[i + 1, j + 1]))
                if i != 0 and j != nc - 1 and nodemap[i - 1, j + 1] != 0:
                    I.append(nodemap[i, j]); J.append(nodemap[i - 1, j + 1]); V.append(f2(gmap[i, j], gmap[i - 1, j + 1]))
    m = int(nodemap.max())
    a = sp.coo_matrix((np.array(V, dtype=np.float64), (np.array(I, dtype=np.int64) - 1, np.array(J, dtype=np.int64) - 1)),
                      shape=(m, m)).tocsr()  # sums duplicates like sparse(I,J,V)
    a = (a + a.T).tocsr()
    a.sort_indices()
    return a


def laplacian(a):
    """core.jl:608-634: off-diagonals negated, stored diagonal zeroed and excluded from the degree."""
    a = a.tocsr().copy()
    n = a.shape[0]
    rows = np.repeat(np.arange(n), np.diff(a.indptr))
    offd = rows != a.indices
    deg = np.bincount(rows[offd], weights=a.data[offd], minlength=n)
    data = np.where(offd, -a.data, 0.0)
    G = sp.csr_matrix((data, a.indices.copy(), a.indptr.copy()), shape=a.shape) + sp.diags(deg, format="csr")
    G = G.tocsr()
    G.sort_indices()
    return G


def connected_components(A):
    """Graphs.connected_components(SimpleGraph(A)): edge where entry != 0; components ordered by their
    smallest vertex, vertices ascending; 1-based ids."""
    A = A.tocsr()
    pat = sp.csr_matrix((A.data != 0).astype(np.int8))
    pat = sp.csr_matrix(((A.data != 0).astype(np.int8), A.indices, A.indptr), shape=A.shape)
    pat.eliminate_zeros()
    ncomp, labels = _scc(pat, directed=False)
    order = {}
    comps = []
    for v, l in enumerate(labels):
        if l not in order:
            order[l] = len(comps)
            comps.append([])
        comps[order[l]].append(v + 1)
    return [np.array(c, dtype=np.int64) for c in comps]


def prune_points(points_rc, point_ids):
    keep = [k for k, p in enumerate(points_rc[2]) if p in point_ids]
    return tuple([lst[k] for k in keep] for lst in points_rc)


def generate_exclude_pairs(points_rc, included_pairs):
    """pairwise.jl:240-269. Returns (exclude list of (id,id) tuples, possibly pruned points_rc)."""
    mat = np.asarray(included_pairs["matrix"])
    ids = included_pairs["point_ids"]
    ex = []
    if included_pairs["mode"] == "include":
        points_rc = prune_points(points_rc, ids)
        for j in range(mat.shape[1]):
            for i in range(mat.shape[0]):
                if mat[i, j] == 0 and mat[j, i] == 0:
                    ex.append((ids[i], ids[j]))
    else:
        for j in range(mat.shape[1]):
            for i in range(mat.shape[0]):
                if mat[i, j] == 1 and mat[j, i] == 1:
                    ex.append((ids[i], ids[j]))
    return ex, points_rc


def compute_graph_data_no_polygons(gmap, polymap, points_rc, included_pairs, avg_res, four_neighbors):
    """pairwise.jl:192-238."""
    gmap = np.asarray(gmap, dtype=np.float64)
    pm = None if polymap is None else np.asarray(polymap, dtype=np.int64)
    nodemap = construct_node_map(gmap, pm)
    G = laplacian(construct_graph(gmap, nodemap, avg_res, four_neighbors))
    cc = connected_components(G)
    if included_pairs is not None:
        exclude, points_rc = generate_exclude_pairs(points_rc, included_pairs)
    else:
        exclude = []
    points = np.array([nodemap[i - 1, j - 1] for i, j in zip(points_rc[0], points_rc[1])], dtype=np.int64)
    return RefProblem(G, cc, points, np.array(points_rc[2], dtype=np.int64), exclude, nodemap, pm, True)


def create_new_polymap(gmap, polymap, points_rc, pt1, pt2):
    """pairwise.jl:369-442, pt1/pt2 branch (point_map argument empty)."""
    pi, pj, pv = points_rc
    f = lambda x: (pi[x] - 1, pj[x] - 1)
    if polymap is None or polymap.size == 0:
        newpoly = np.zeros(gmap.shape, dtype=np.int64)
        for x in [k for k, v in enumerate(pv) if v == pt1]:
            newpoly[f(x)] = pt1
        for x in [k for k, v in enumerate(pv) if v == pt2]:
            newpoly[f(x)] = pt2
        return newpoly
    newpoly = polymap.copy()
    k = int(polymap.max())
    for p in (pt1, pt2):
        idx = [q for q, v in enumerate(pv) if v == p]
        if len(idx) == 1:
            continue
        allzero = all(polymap[f(x)] == 0 for x in idx)
        if allzero:
            for x in idx:
                newpoly[f(x)] = k + 1
            k += 1
        else:
            nz = [x for x in idx if polymap[f(x)] != 0]
            if len(nz) == 1:
                # pairwise.jl:424 references an undefined variable (`overlap`) on this branch; the
                # reference would throw here, so no fixture can exercise it.
                raise NotImplementedError("reference branch pairwise.jl:424 is not executable")
            vals = [polymap[f(x)] for x in nz]
            newpoly[np.isin(polymap, vals)] = k + 1
            k += 1
    return newpoly


def compute_graph_data_polygons(gmap, polymap, points_rc, pt1, pt2, avg_res, four_neighbors):
    """pairwise.jl:137-190: a fresh graph for ONE pair of focal regions."""
    gmap = np.asarray(gmap, dtype=np.float64)
    pm = None if polymap is None else np.asarray(polymap, dtype=np.int64)
    newpoly = create_new_polymap(gmap, pm, points_rc, pt1, pt2)
    nodemap = construct_node_map(gmap, newpoly)
    a = construct_graph(gmap, nodemap, avg_res, four_neighbors)
    G = laplacian(a)
    offd = a.copy().tolil()
    offd.setdiag(0)
    cc = connected_components(offd.tocsr())
    x = list(points_rc[2]).index(pt1)
    y = list(points_rc[2]).index(pt2)
    c1 = nodemap[points_rc[0][x] - 1, points_rc[1][x] - 1]
    c2 = nodemap[points_rc[0][y] - 1, points_rc[1][y] - 1]
    return RefProblem(G, cc, np.array([c1, c2], dtype=np.int64), np.array([pt1, pt2], dtype=np.int64), [], nodemap,
                      newpoly, True)


def compute_graph_data_network(ei, ej, ev, focal):
    """network/pairwise.jl:31-65."""
    ei = np.asarray(ei, dtype=np.int64)
    ej = np.asarray(ej, dtype=np.int64)
    m = int(max(ei.max(), ej.max()))
    A = sp.coo_matrix((np.asarray(ev, dtype=np.float64), (ei - 1, ej - 1)), shape=(m, m)).tocsr()
    A = (A + A.T).tocsr()
    A.sort_indices()
    cc = connected_components(A)
    G = laplacian(A)
    fp = np.asarray(focal, dtype=np.int64)
    return RefProblem(G, cc, fp, fp, [], None, None, False)


def synthetic_raster_problem(nrows, ncols, sigma=1.0, seed=12345, four_neighbors=False, avg_res=False):
    """SURVEY.md section 8(d) synthetic raster: all cells valid, r = exp(sigma*N(0,1)), g = 1/r.
    Vectorised equivalent of construct_node_map + construct_graph + laplacian! for an all-valid raster
    without polygons (column-major node numbering, pairwise.jl:273-275,327-353). Returns scipy CSR Laplacian."""
    rng = np.random.default_rng(seed)
    r = np.exp(sigma * rng.standard_normal((nrows, ncols)))
    g = 1.0 / r
    return raster_laplacian_from_conductance(g, four_neighbors, avg_res), g


def raster_laplacian_from_conductance(g, four_neighbors=False, avg_res=False):
    nrows, ncols = g.shape
    node = (np.arange(nrows * ncols, dtype=np.int64).reshape(ncols, nrows).T)  # column-major ids, 0-based

    def avg(x, y, diag):
        if avg_res:
            v = 1.0 / ((1.0 / x + 1.0 / y) / 2.0)
            return v / SQRT2 if diag else v
        v = (x + y) / 2.0
        return v / SQRT2 if diag else v

    I, J, V = [], [], []
    # E neighbour
    I.append(node[:, :-1].ravel()); J.append(node[:, 1:].ravel()); V.append(avg(g[:, :-1], g[:, 1:], False).ravel())
    # S neighbour
    I.append(node[:-1, :].ravel()); J.append(node[1:, :].ravel()); V.append(avg(g[:-1, :], g[1:, :], False).ravel())
    if not four_neighbors:
        I.append(node[:-1, :-1].ravel()); J.append(node[1:, 1:].ravel()); V.append(avg(g[:-1, :-1], g[1:, 1:], True).ravel())
        I.append(node[1:, :-1].ravel()); J.append(node[:-1, 1:].ravel()); V.append(avg(g[1:, :-1], g[:-1, 1:], True).ravel())
    I = np.concatenate(I); J = np.concatenate(J); V = np.concatenate(V)
    n = nrows * ncols
    a = sp.coo_matrix((V, (I, J)), shape=(n, n)).tocsr()
    a = (a + a.T).tocsr()
    deg = np.asarray(a.sum(axis=1)).ravel()
    G = (sp.diags(deg, format="csr") - a).tocsr()
    G.sort_indices()
    return G
