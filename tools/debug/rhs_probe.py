"""GPU probe: which call of check_cellspace's sequence fails, under which options."""
import os, sys, json, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import circuitscape_jl_amd
from circuitscape_jl_amd import lib as L
L.load(os.environ.get("CSGPU_LIB"))
from helpers import _nodata_raster
shape = tuple(int(v) for v in os.environ.get("SHAPE", "310,287").split(",")); batch = 8
for four in (False, True):
    g = _nodata_raster(shape, 7 + four)
    for mode in ("cell", "compact"):
        if mode == "compact": os.environ["CSGPU_NO_CELLSPACE"] = "1"
        else: os.environ.pop("CSGPU_NO_CELLSPACE", None)
        for pb in (0, 4):
            for variant in ("seq", "seq_nograph", "rhs_first"):
                o = L.default_opts(batch=batch, precond_bytes=pb, itmax=400, use_graph=-1 if variant == "seq_nograph" else 0)
                with L.raster_setup(g, o, four_neighbors=four) as h:
                    n = h.info["n"]
                    labels, nc = h.components()
                    big = np.flatnonzero(labels == np.bincount(labels).argmax())
                    ids = np.random.default_rng(5).choice(big, size=2 * batch + 2, replace=False)
                    src = [int(v) for v in ids[:batch + 1]]; dst = [int(v) for v in ids[batch + 1:]]
                    rng = np.random.default_rng(9)
                    B = rng.standard_normal((n, 3)); B -= B.mean(axis=0)
                    for c in range(nc):
                        m = labels == c
                        B[m] -= B[m].mean(axis=0)
                    res = {}
                    def tryit(name, f):
                        try:
                            st = f()
                            res[name] = (st["total_iters"], float("%.2e" % st["max_relres"]))
                        except Exception as e:
                            res[name] = "FAIL " + str(e)[:90]
                    if variant != "rhs_first":
                        tryit("pairs_v", lambda: h.solve_pairs(src, dst, gather=ids[:5], want_voltages=True)[3])
                        tryit("pairs", lambda: h.solve_pairs(src, dst)[3])
                        cum = np.zeros(n); mx = np.zeros(n)
                        tryit("cur", lambda: h.solve_pairs_currents(src[:3], dst[:3], cum=cum, mx=mx)[3])
                    tryit("rhs", lambda: h.solve_rhs(B)[1])
                    tryit("rhs2", lambda: h.solve_rhs(B)[1])
                    tryit("grounded", lambda: h.solve_grounded(B[:, :2], [[src[0], dst[0]], [src[1]]], want_currents=True)[2])
                    print(json.dumps({"four": four, "mode": mode, "pb": pb, "variant": variant, "nc": int(nc), **{k: v for k, v in res.items()}}), flush=True)
