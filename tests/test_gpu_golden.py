"""The reference's own regression cases (test/test_utils.jl:77-89,101-114) through the real HIP path:
golden fixture -> GraphProblem -> host mirror of solve(prob, ::HIPAMGSolver, ...) -> C ABI -> MI355X."""
import numpy as np
import pytest

from conftest import compare_resistances, golden_cases, load_case
from helpers import expected_ids, run_fixture

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", golden_cases())
def test_golden_fixture_on_gpu(gpu_lib, name):
    from circuitscape_jl_amd import solver as ps
    import hostmirror as hm
    case = load_case(name)
    got = run_fixture(case, ps.HIPAMGSolver(bs=8))
    exp = np.array(case["expected"])
    assert np.array_equal(expected_ids(case), got[1:, 0])
    # reference tolerance: |x - r| <= 1e-3 (test_utils.jl:72-73,147); held here to 1e-6 relative
    compare_resistances(exp[1:, 1:], got[1:, 1:], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("name", ["sgVerify12", "sgVerify4", "sgNetworkVerify1"])
def test_golden_fixture_fp32_preconditioner(gpu_lib, name):
    from circuitscape_jl_amd import solver as ps
    import hostmirror as hm
    case = load_case(name)
    got = run_fixture(case, ps.HIPAMGSolver(bs=4, opts={"precond_bytes": 4}))
    exp = np.array(case["expected"])
    compare_resistances(exp[1:, 1:], got[1:, 1:], rtol=1e-6, atol=1e-9)


def test_fp32_preconditioner_matches_tight_oracle(gpu_lib, oracle):
    from oracle import refgraph as rg
    N = 400
    G, g = rg.synthetic_raster_problem(N, N)
    A = oracle.regularize(G)
    cells = np.random.default_rng(67890).choice(N * N, size=5, replace=False)
    src = [cells[i] for i in range(5) for j in range(i + 1, 5)]
    dst = [cells[j] for i in range(5) for j in range(i + 1, 5)]
    Ro, _, _ = oracle.OracleAMG(A).solve_pairs(src, dst, rtol=1e-12, atol=0.0, criterion=1)
    for pb in (0, 4):
        h = gpu_lib.raster_setup(g, gpu_lib.default_opts(batch=8, precond_bytes=pb))
        R, _, _, st = h.solve_pairs(src, dst)
        assert st["not_converged"] == 0 and st["max_relres"] < 1e-4
        assert np.max(np.abs(R - Ro) / Ro) < 1e-6, pb
        h.close()


def test_single_precision_handle(gpu_lib, oracle):
    """val_bytes = 4 (the reference's precision = single): everything fp32. The reference's own single-precision
    tolerance is 1e-2 absolute (test_utils.jl:72-73); we state and hold 1e-3 relative against the fp64 oracle
    (the eps(Float32)*norm(nzval) regularisation alone perturbs the conductances at the 1e-4 level)."""
    from oracle import refgraph as rg
    N = 200
    G, g = rg.synthetic_raster_problem(N, N)
    A64 = oracle.regularize(G)
    cells = np.random.default_rng(3).choice(N * N, size=4, replace=False)
    src, dst = [cells[0], cells[1], cells[2]], [cells[1], cells[2], cells[3]]
    Ro, _, _ = oracle.OracleAMG(A64).solve_pairs(src, dst, rtol=1e-12, atol=0.0, criterion=1)
    h = gpu_lib.raster_setup(g.astype(np.float32), gpu_lib.default_opts(batch=4, rtol=1e-5, atol=0.0), reg=False)
    R, _, _, st = h.solve_pairs(src, dst)
    assert R.dtype == np.float32
    assert np.max(np.abs(R - Ro) / Ro) < 1e-3
    h.close()


def test_not_converged_is_reported_like_the_reference(gpu_lib, oracle):
    """itmax exhausted -> status 1 and the reference's error wording (core.jl:641)."""
    from oracle import refgraph as rg
    from circuitscape_jl_amd import solver as ps
    import hostmirror as hm
    G, g = rg.synthetic_raster_problem(100, 100)
    h = gpu_lib.raster_setup(g, gpu_lib.default_opts(itmax=1, batch=1))
    with pytest.raises(gpu_lib.CsgpuError) as e:
        h.solve_pairs([0], [9999])
    assert e.value.code == gpu_lib.CSGPU_NOT_CONVERGED and "did not converge" in str(e.value)
    h.close()


@pytest.mark.parametrize("name", __import__("conftest").advanced_cases())
def test_network_advanced_on_gpu(gpu_lib, name):
    from circuitscape_jl_amd import solver as ps
    import hostmirror as hm
    from helpers import run_network_advanced_fixture
    case = load_case(name)
    got = run_network_advanced_fixture(case, ps.HIPAMGSolver(bs=1))
    exp = np.array(case["expected_voltages"])
    assert np.array_equal(exp[:, 0] + 1, got[:, 0])
    assert np.max(np.abs(exp[:, 1] - got[:, 1])) <= 1e-5 * max(1.0, np.abs(exp[:, 1]).max())


@pytest.mark.parametrize("name", ["sgVerify1", "sgVerify3", "sgVerify4", "sgVerify5", "sgVerify9", "sgVerify11",
                                  "sgVerify13", "sgVerify14"])
def test_current_and_voltage_maps_on_gpu(gpu_lib, name):
    """scope row N1 on the real device: node currents / cumulative / maximum maps computed by the HIP kernels in
    csrc/currents.h, compared with the reference's golden .asc maps (criterion sum(abs2, x - r) < 1e-6)."""
    from circuitscape_jl_amd import solver as ps
    import hostmirror as hm
    from test_emu_solver import _check_maps
    case = load_case(name)
    st = {}
    run_fixture(case, ps.HIPAMGSolver(bs=4, opts={"rtol": 1e-10, "atol": 0.0, "criterion": 1}), stats=st)
    ncmp = _check_maps(case, st)
    if name != "sgVerify3":
        assert ncmp > 0


def test_node_currents_conserve_charge_large(gpu_lib):
    """Size-independent property of the current kernel at a size the oracle is not run: with a unit current injected
    at src and extracted at dst, the node current is 1 at both terminals and Kirchhoff holds elsewhere (in == out), so
    the sum over the cut between a terminal and the rest is 1."""
    N = 1200
    g = 1.0 / np.exp(np.random.default_rng(12345).standard_normal((N, N)))
    h = gpu_lib.raster_setup(g, gpu_lib.default_opts(batch=2, criterion=gpu_lib.CRIT_TRUE_RESIDUAL, rtol=1e-10, atol=0.0))
    src, dst = [5 * N + 7, 100], [900 * N + 650, N * N - 3]
    cum = np.zeros(N * N)
    R, V, C, st = h.solve_pairs_currents(src, dst, want_voltages=False, want_currents=True, cum=cum)
    for p in range(2):
        assert abs(C[src[p], p] - 1.0) < 1e-6 and abs(C[dst[p], p] - 1.0) < 1e-6
        assert C[:, p].min() >= 0.0
    assert np.max(np.abs(cum - C.sum(axis=1))) < 1e-9
    h.close()


@pytest.mark.parametrize("name", ["sgNetworkVerify1", "sgNetworkVerify2", "sgNetworkVerify3"])
def test_network_current_tables_on_gpu(gpu_lib, name):
    from circuitscape_jl_amd import solver as ps
    import hostmirror as hm
    from test_emu_solver import _check_network_tables
    case = load_case(name)
    st = {"want_tables": True}
    run_fixture(case, ps.HIPAMGSolver(bs=4, opts={"rtol": 1e-10, "atol": 0.0, "criterion": 1}), stats=st)
    assert _check_network_tables(case, st) > 0


@pytest.mark.parametrize("name", __import__("conftest").raster_advanced_cases())
def test_raster_advanced_on_gpu(gpu_lib, name):
    """scope row N2: raster advanced mode (mgVerify1..6) on the device with the reference's stopping rule; maps
    against the goldens with the reference's criterion."""
    from circuitscape_jl_amd import solver as ps
    import hostmirror as hm
    from conftest import compare_aagrid, load_case
    from helpers import run_raster_advanced_fixture
    case = load_case(name)
    _, _, maps = run_raster_advanced_fixture(case, ps.HIPAMGSolver(bs=1))
    for key, exp in case["expected"].items():
        assert compare_aagrid(exp, maps[key]), (name, key)


@pytest.mark.parametrize("name", __import__("conftest").onetoall_cases())
def test_onetoall_alltoone_on_gpu(gpu_lib, name):
    """scope row N2: the 25 one-to-all / all-to-one cases on the device with the reference's stopping rule."""
    from circuitscape_jl_amd import solver as ps
    import hostmirror as hm
    from conftest import load_case
    from helpers import check_onetoall_against_golden, run_onetoall_fixture
    case = load_case(name)
    res, cum, pts = run_onetoall_fixture(case, ps.HIPAMGSolver(bs=1))
    check_onetoall_against_golden(case, res, cum, pts)


def test_compute_omniscape_current_on_gpu(gpu_lib):
    """scope row N3 (entry point only): a 201-cell-wide circular moving window through compute_omniscape_current on
    the device (reference stopping rule) against the oracle's direct solve of the same grounded system."""
    from circuitscape_jl_amd import solver as ps
    import hostmirror as hm
    from helpers import _build_graph
    from oracle import refmaps
    from test_emu_solver import _omniscape_window
    cond, src, gnd = _omniscape_window(201, 5)
    build = _build_graph({"connect_using_avg_resistances": False, "connect_four_neighbors_only": False})
    got = hm.compute_omniscape_current(cond, src, gnd, {"solver": "hip"}, build, solver=ps.HIPAMGSolver(bs=1))
    ref = refmaps.compute_omniscape_current(cond, src, gnd, four_neighbors=False, mode="direct")
    assert np.max(np.abs(got - ref)) < 2e-5 * ref.max()
    assert abs(got[gnd > 0].sum() - src[cond > 0].sum()) < 1e-4 * src.sum()


@pytest.mark.parametrize("name", ["sgVerify4", "sgVerify13", "sgVerify17"])
def test_raster_pairwise_with_device_built_graph_on_gpu(gpu_lib, name):
    """scope row N4 end to end on the device: graph layer (node map, Laplacian, regularisation, components) and all
    pair solves on one handle, reference stopping rule, against the golden resistances."""
    from circuitscape_jl_amd import solver as ps
    import hostmirror as hm
    from oracle import refgraph as rg
    case = load_case(name)
    o = case["options"]
    points_rc = tuple(list(x) for x in case["points_rc"])
    exclude = []
    if case["included_pairs"] is not None:
        exclude, points_rc = rg.generate_exclude_pairs(points_rc, case["included_pairs"])
    got = ps.raster_pairwise_on_device(np.array(case["cellmap"], dtype=np.float64), points_rc, ps.HIPAMGSolver(bs=8),
                                       four_neighbors=o["connect_four_neighbors_only"],
                                       avg_res=o["connect_using_avg_resistances"], exclude_pairs=exclude)
    exp = np.array(case["expected"])
    assert np.array_equal(exp[1:, 0], got[1:, 0])
    compare_resistances(exp[1:, 1:], got[1:, 1:], rtol=1e-6, atol=1e-9)


def test_omniscape_batch_on_gpu(gpu_lib):
    """scope row N3: 12 moving windows stacked into one raster, one block-diagonal PCG on the device (reference
    stopping rule + polishing); every window against the oracle's direct solve."""
    from circuitscape_jl_amd import solver as ps
    import hostmirror as hm
    from oracle import refmaps
    from test_emu_solver import _omniscape_window
    wins = [_omniscape_window(61 + 10 * (k % 4), 100 + k) for k in range(12)]
    maps, st = ps.compute_omniscape_current_batch(wins, {"connect_four_neighbors_only": "False"},
                                                  solver=ps.HIPAMGSolver(bs=1))
    assert st["not_converged"] == 0
    for (cond, src, gnd), got in zip(wins, maps):
        ref = refmaps.compute_omniscape_current(cond, src, gnd, four_neighbors=False, mode="direct")
        assert np.max(np.abs(got - ref)) < 5e-5 * ref.max()


@pytest.mark.parametrize("name", ["mgVerify2", "mgVerify6"])
def test_raster_advanced_on_device_with_direct_grounds_on_gpu(gpu_lib, name):
    """scope rows N2 + N4 on the device: polygon-free raster advanced cases (direct grounds) through
    csgpu_raster_setup_grounded + csgpu_solve_raster, reference stopping rule, golden maps."""
    from circuitscape_jl_amd import solver as ps
    import hostmirror as hm
    from conftest import compare_aagrid
    from helpers import _float_map, flags_from_case
    case = load_case(name)
    o = case["options"]
    flags = flags_from_case(case, True)
    flags.policy = o["remove_src_or_gnd"]
    vm, cm = ps.raster_advanced_on_device(np.array(case["cellmap"]), _float_map(case["source_map"]),
                                          _float_map(case["ground_map"]), flags, ps.HIPAMGSolver(bs=1),
                                          four_neighbors=o["connect_four_neighbors_only"],
                                          avg_res=o["connect_using_avg_resistances"])
    got = {"voltmap": vm, "curmap": cm}
    for key, exp in case["expected"].items():
        assert compare_aagrid(exp, got[key]), (name, key)


@pytest.mark.parametrize("name", ["oneToAllVerify4", "allToOneVerify4"])
def test_onetoall_on_device_built_graph_on_gpu(gpu_lib, name):
    """scope rows N2 + N4 on the device: polygon-free one-to-all / all-to-one with single-cell focal points, every
    per-point solve on the device-built graph; golden resistances and maps."""
    from circuitscape_jl_amd import solver as ps
    import hostmirror as hm
    from helpers import check_onetoall_against_golden, flags_from_case
    case = load_case(name)
    o = case["options"]
    flags = flags_from_case(case, True)
    flags.is_onetoall = case["kind"] == "one_to_all"
    flags.is_alltoone = not flags.is_onetoall
    res, cum, pts = ps.onetoall_on_device(np.array(case["cellmap"], dtype=np.float64), case["points_rc"], flags,
                                          ps.HIPAMGSolver(bs=1), four_neighbors=o["connect_four_neighbors_only"],
                                          avg_res=o["connect_using_avg_resistances"])
    assert check_onetoall_against_golden(case, res, cum, pts) > 0


def test_polygon_graph_built_on_device(gpu_lib):
    from helpers import check_polygon_graph_on_device
    check_polygon_graph_on_device(gpu_lib)


@pytest.mark.parametrize("name", [c for c in golden_cases() if not c.startswith("sgNetwork")])
def test_every_raster_pairwise_golden_with_device_built_graph(gpu_lib, name):
    """All 17 raster pairwise cases of the reference with the graph layer (node numbering, polygon merge with summed
    parallel edges, Laplacian, components) on the device, default tolerances, against the golden resistances."""
    from circuitscape_jl_amd import solver as ps
    from helpers import run_fixture_device_graph
    case = load_case(name)
    got = run_fixture_device_graph(case, ps.HIPAMGSolver(bs=8))
    exp = np.array(case["expected"])
    assert np.array_equal(exp[1:, 0], got[1:, 0])
    compare_resistances(exp[1:, 1:], got[1:, 1:], rtol=1e-6, atol=1e-9)


def test_omniscape_moving_window_driver_on_gpu(gpu_lib):
    """scope row N3: solver.omniscape_moving_window (windows -> block-diagonal device solves -> mosaic) on the device
    against the checker (every window through the oracle's compute_omniscape_current, direct solves)."""
    from circuitscape_jl_amd import solver as ps
    from oracle import refmaps
    from test_emu_solver import _omniscape_landscape
    cond, strength = _omniscape_landscape((64, 57), 9)
    ref, nref = refmaps.omniscape_moving_window(cond, strength, radius=12, block_size=5)
    tight = ps.HIPAMGSolver(bs=1, opts={"rtol": 1e-10, "atol": 0.0, "criterion": 1})
    got, nwin = ps.omniscape_moving_window(cond, strength, radius=12, block_size=5, solver=tight, windows_per_solve=48)
    assert nwin == nref and nref > 100
    assert np.all(got[cond == 0] == 0)
    assert np.max(np.abs(got - ref)) < 1e-7 * ref.max()
    # the reference's default tolerances (rtol 1e-6 on the M-norm, 1e-4 post-check per component)
    got2, _ = ps.omniscape_moving_window(cond, strength, radius=12, block_size=5, solver=ps.HIPAMGSolver(bs=1))
    assert np.max(np.abs(got2 - ref)) < 1e-3 * ref.max()


@pytest.mark.parametrize("name", ["oneToAllVerify4", "allToOneVerify4"])
def test_onetoall_cumulative_maps_through_sparse_sources_on_gpu(gpu_lib, name):
    """csgpu_solve_sources pinned on the reference's goldens (see helpers.check_onetoall_sparse_sources_against_golden)."""
    from helpers import check_onetoall_sparse_sources_against_golden
    check_onetoall_sparse_sources_against_golden(name)
