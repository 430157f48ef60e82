#!/usr/bin/env python3
"""One-to-all on a synthetic raster: per-point re-setup (what the reference does: a fresh smoothed_aggregation per focal
point, src/raster/onetoall.jl:106-151 -> advanced.jl:307-312; here csgpu_raster_setup_grounded + csgpu_solve_raster per
point) against ONE setup + csgpu_solve_grounded batches on the shared hierarchy. Prints one JSON line per size."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import circuitscape_jl_amd  # noqa
from circuitscape_jl_amd import lib, solver as ps

lib.load(os.environ.get("CSGPU_LIB"))
for arg in (sys.argv[1:] or ["2000"]):
    N = int(arg)
    npts = int(os.environ.get("NPTS", "16"))
    g = bench.make_raster(N)
    rng = np.random.default_rng(4242)
    cells = rng.choice(N * N, size=npts, replace=False)
    rows, cols = cells // N + 1, cells % N + 1
    points_rc = (rows, cols, np.arange(1, npts + 1))
    flags = ps.Flags(is_raster=True, outputflags=ps.OutputFlags(write_cur_maps=True), policy="rmvgnd")
    flags.is_onetoall, flags.is_alltoone = True, False
    sv = ps.HIPAMGSolver(bs=16, opts={"precond_bytes": 4})
    # warm-up (code objects, pool)
    ps.onetoall_on_device(g[:256, :256], (rows % 256 + 1, cols % 256 + 1, np.arange(1, npts + 1)), flags, sv)
    st = {}
    t0 = time.perf_counter()
    res, cum, pts = ps.onetoall_on_device(g, points_rc, flags, sv, stats=st)
    t_shared = time.perf_counter() - t0
    # per-point path: graph build + setup + solve + current map per focal point
    point_map = np.zeros(g.shape, dtype=np.int64)
    point_map[rows - 1, cols - 1] = np.arange(1, npts + 1)
    raw = ps.OutputFlags()
    t0 = time.perf_counter()
    res2 = np.zeros(npts)
    nper = min(npts, int(os.environ.get("NPER", "4")))
    for i in range(nper):
        me = point_map == i + 1
        others = (point_map != 0) & ~me
        sub = ps.Flags(is_raster=True, outputflags=raw, policy="rmvgnd")
        vol, cur = ps.raster_advanced_on_device(g, np.where(me, 1.0, 0.0), np.where(others, np.inf, 0.0), sub,
                                                ps.HIPAMGSolver(bs=1, opts={"precond_bytes": 4}))
        res2[i] = vol[rows[i] - 1, cols[i] - 1]
    t_per = (time.perf_counter() - t0) / nper * npts
    out = dict(size=N, points=npts, shared_hierarchy_s=t_shared, per_point_setup_s_extrapolated=t_per, per_point_timed=nper,
               speedup=t_per / t_shared, iters_mean_shared=st.get("total_iters", 0) / npts,
               max_rel_diff_resistance=float(np.max(np.abs(res[:nper, 1] - res2[:nper]) / np.abs(res2[:nper]))))
    print(json.dumps(out), flush=True)
