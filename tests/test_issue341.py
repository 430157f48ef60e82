"""The reference's include / exclude-pairs scenarios (test/issue341.jl, run by test/runtests.jl:43-45 with solver = cg+amg)
mirrored through the product path (solver.py::solve behind single_ground_all_pairs, src/core.jl:96-305) and through the
oracle's host restatement: an include list restricts the solves and prunes the other focal points
(src/raster/pairwise.jl:240-269), excluded pairs stay -1, a non-empty exclude list switches the resistance shortcut off
(src/core.jl:140), focal REGIONS take the per-pair polygon path (raster/pairwise.jl:72-135). The two existing-file
scenarios of that test (sgVerify13, sgVerify17) are golden fixtures already (tests/test_emu_solver.py, test_gpu_golden.py).
CPU: emulator build; `-m gpu`: the same scenarios on the device."""
import numpy as np
import pytest

from helpers import run_fixture

OPTS = {"connect_four_neighbors_only": True, "connect_using_avg_resistances": True, "use_polygons": False, "use_mask": False,
        "use_included_pairs": True, "write_volt_maps": False, "write_cur_maps": False, "write_cum_cur_map_only": False,
        "write_max_cur_maps": False, "log_transform_maps": False, "set_null_currents_to_nodata": False,
        "set_null_voltages_to_nodata": False}


def _pairs_table(mode, pairs):
    """read_included_pairs of a `mode include|exclude` text file (src/io.jl:328-385), as tests/golden/make_golden.py does it"""
    ids = sorted({v for p in pairs for v in p})
    mat = np.zeros((len(ids), len(ids)), dtype=int)
    for a, b in pairs:
        mat[ids.index(a), ids.index(b)] = mat[ids.index(b), ids.index(a)] = 1
    return {"mode": mode, "point_ids": ids, "matrix": mat.tolist()}


def _case(n, pts, mode, pairs):
    """n x n raster of unit resistances (habitat_map_is_resistances = True -> conductance 1), focal cells `pts` =
    {(row, col): id} (0-based), an include / exclude list."""
    rows, cols, ids = [], [], []
    for j in range(n):                       # column-major order, as findall on the Julia matrix returns them ...
        for i in range(n):
            if (i, j) in pts:
                rows.append(i + 1)
                cols.append(j + 1)
                ids.append(pts[(i, j)])
    order = np.argsort(ids, kind="stable")   # ... then sorted by id (read_point_map, src/io.jl:196-249)
    points_rc = [[rows[k] for k in order], [cols[k] for k in order], [ids[k] for k in order]]
    return {"name": "issue341", "kind": "raster", "options": dict(OPTS), "cellmap": np.ones((n, n)).tolist(), "polymap": None,
            "points_rc": points_rc, "included_pairs": _pairs_table(mode, pairs)}


SCENARIOS = {
    # Test 1: three focal points, include only (1, 2): point 3 is pruned, result 3 x 3
    "include_prunes": (_case(5, {(0, 0): 1, (0, 4): 2, (4, 0): 3}, "include", [(1, 2)]), [1, 2], [(1, 2)], []),
    # Test 4: focal regions (two cells per id), include only (1, 2)
    "include_regions": (_case(6, {(0, 0): 1, (0, 1): 1, (0, 4): 2, (0, 5): 2, (5, 0): 3}, "include", [(1, 2)]), [1, 2], [(1, 2)], []),
    # Test 5: exclude (1, 3): (1, 2) and (2, 3) solved, (1, 3) stays -1
    "exclude_one": (_case(5, {(0, 0): 1, (0, 4): 2, (4, 0): 3}, "exclude", [(1, 3)]), [1, 2, 3], [(1, 2), (2, 3)], [(1, 3)]),
    # Test 6: exclude (1, 3) and (2, 4)
    "exclude_two": (_case(5, {(0, 0): 1, (0, 4): 2, (4, 0): 3, (4, 4): 4}, "exclude", [(1, 3), (2, 4)]), [1, 2, 3, 4],
                    [(1, 2), (1, 4), (2, 3), (3, 4)], [(1, 3), (2, 4)]),
    # Test 7: exclude with focal regions (polygon path)
    "exclude_regions": (_case(6, {(0, 0): 1, (0, 1): 1, (0, 4): 2, (0, 5): 2, (5, 0): 3}, "exclude", [(1, 3)]), [1, 2, 3],
                        [(1, 2), (2, 3)], [(1, 3)]),
}


def _check(lib, oracle, name):
    from circuitscape_jl_amd import solver as ps
    case, ids, solved, excluded = SCENARIOS[name]
    st = {}
    got = run_fixture(case, ps.HIPAMGSolver(bs=4), stats=st)
    want = oracle.raster_pairwise_from_fixture(case, mode="direct")
    assert got.shape == (len(ids) + 1, len(ids) + 1) and want.shape == got.shape          # pruned points are gone
    assert list(got[0, 1:]) == ids and list(got[1:, 0]) == ids and got[0, 0] == 0
    pos = {v: k + 1 for k, v in enumerate(ids)}
    for a, b in solved:
        assert got[pos[a], pos[b]] > 0 and got[pos[b], pos[a]] == got[pos[a], pos[b]]
        assert abs(got[pos[a], pos[b]] - want[pos[a], pos[b]]) <= 1e-6 * want[pos[a], pos[b]]
    for a, b in excluded:
        assert got[pos[a], pos[b]] == -1 and got[pos[b], pos[a]] == -1 and want[pos[a], pos[b]] == -1
    assert np.all(np.diag(got)[1:] == 0)
    if "regions" not in name:
        # a non-empty exclude list disables the shortcut (core.jl:140) -- and an include list always leaves one: the
        # reference's generate_exclude_pairs pushes the (i, i) entries of the table too (raster/pairwise.jl:249-255)
        assert st["shortcut"] is False
        assert st["nsolves"] == len(solved)


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_issue341_scenarios(emu_lib, oracle, name):
    _check(emu_lib, oracle, name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_issue341_scenarios_gpu(gpu_lib, oracle, name):
    _check(gpu_lib, oracle, name)
