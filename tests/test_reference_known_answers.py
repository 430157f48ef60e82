"""Known-answer vectors of the reference's own unit tests (test/internal.jl:44-175) against the oracle's graph
construction (oracle/refgraph.py) and the host-mirror helpers of the product (circuitscape.jl_amd/solver.py)."""
import numpy as np
import pytest

import circuitscape_jl_amd  # noqa: F401
from circuitscape_jl_amd import solver as ps
import hostmirror as hm
from oracle import refgraph as rg
from oracle import refonetoall, refsolve

NODE_MAP_CASES = [  # (gmap, polymap or None, expected nodemap)  test/internal.jl:45-102
    ([[0, 1, 2], [2, 0, 0], [2, 0, 2]], None, [[0, 3, 4], [1, 0, 0], [2, 0, 5]]),
    ([[0, 1, 2], [2, 0, 0], [2, 0, 2]], [[1, 0, 1], [2, 1, 0], [0, 0, 2]], [[4, 3, 4], [1, 4, 0], [2, 0, 1]]),
    ([[1, 0, 1], [0, 1, 0], [1, 0, 1]], [[1, 0, 1], [0, 2, 0], [2, 0, 0]], [[1, 0, 1], [0, 2, 0], [2, 0, 3]]),
    ([[0, 0, 0, 1.0, 1.0], [0, 0, 0, 3.01, 2.0], [1.0, 2.0, 2.0, 1.0, 1.0], [1.0, 2.0, 2.0, 1.0, 1.0],
      [1.0, 2.0, 2.0, 0, 1.0]],
     [[1, 2, 0, 0, 0], [0, 0, 0, 0, 0], [0, 0, 0, 0, 0], [0, 0, 0, 0, 0], [1, 0, 0, 0, 2]],
     [[3, 18, 0, 10, 14], [0, 0, 0, 11, 15], [1, 4, 7, 12, 16], [2, 5, 8, 13, 17], [3, 6, 9, 0, 18]]),
]


@pytest.mark.parametrize("gmap,polymap,expected", NODE_MAP_CASES)
def test_construct_node_map(gmap, polymap, expected):
    g = np.array(gmap, dtype=np.float64)
    pm = None if polymap is None else np.array(polymap, dtype=np.int64)
    assert np.array_equal(rg.construct_node_map(g, pm), np.array(expected))
    assert np.array_equal(hm._construct_node_map(g, pm), np.array(expected))


def test_create_new_polymap_point_map_branch():
    """test/internal.jl:104-126 (oneToAllVerify11's rasters: no cell map value is needed by this branch)."""
    from conftest import load_case
    case = load_case("oneToAllVerify11")
    polymap = np.array(case["polymap"], dtype=np.int64)
    point_map = np.array([[1, 2, 0, 0, 0], [0, 0, 0, 0, 0], [3, 0, 0, 7, 0], [4, 0, 0, 0, 0], [1, 0, 0, 0, 2]])
    expected = np.array([[1, 2, 0, 0, 0], [0, 0, 0, 0, 0], [12, 0, 0, 2, 0], [1, 0, 0, 0, 0], [1, 0, 0, 0, 2]])
    assert np.array_equal(refonetoall.create_new_polymap_pointmap(polymap, case["points_rc"], point_map), expected)
    assert np.array_equal(hm.create_new_polymap(polymap, case["points_rc"], point_map), expected)


@pytest.mark.parametrize("policy,expected", [  # test/internal.jl:130-133
    ("rmvgnd", ([1, 0, 0], [0, 0, 0], [1, 0, 0])), ("rmvsrc", ([0, 0, 0], [1, 0, 0], [1, 0, 0])),
    ("keepall", ([1, 0, 0], [1, 0, 0], [1, 0, 0])), ("rmvall", ([0, 0, 0], [1, 0, 0], [1, 0, 0]))])
def test_resolve_conflicts(policy, expected):
    for f in (hm.resolve_conflicts, refsolve.resolve_conflicts):
        got = f([1.0, 0.0, 0.0], [1.0, 0.0, 0.0], policy)
        for a, b in zip(got, expected):
            assert np.array_equal(np.asarray(a), np.asarray(b, dtype=float))


def test_construct_graph():
    """test/internal.jl:136-171: (avg_res, four_neighbors) -> adjacency, tolerance of the reference's test."""
    gmap = np.array([[0, 1, 2], [2, 0, 0], [2, 0, 2]], dtype=np.float64)
    nodemap = np.array([[0, 3, 4], [1, 0, 0], [2, 0, 5]])

    def dense(pairs):
        m = np.zeros((5, 5))
        for i, j, v in pairs:
            m[i, j] = m[j, i] = v
        return m
    cases = [((False, True), dense([(0, 1, 2), (2, 3, 1.5)])), ((True, True), dense([(0, 1, 2), (2, 3, 1.33333)])),
             ((False, False), dense([(0, 1, 2), (0, 2, 1.06066), (2, 3, 1.5)])),
             ((True, False), dense([(0, 1, 2), (0, 2, 0.942809), (2, 3, 1.3333)]))]
    for (avg_res, four), exp in cases:
        A = rg.construct_graph(gmap, nodemap, avg_res, four)
        assert np.sum((A.toarray() - exp) ** 2) < 1e-6
