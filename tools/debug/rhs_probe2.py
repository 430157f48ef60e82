"""GPU probe 2: four-neighbour cell-space raster, fp32 hierarchy, general rhs: hierarchy fingerprint + A/B knobs (one process per knob)."""
import os, sys, json, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import circuitscape_jl_amd
from circuitscape_jl_amd import lib as L
L.load(os.environ.get("CSGPU_LIB"))
from helpers import _nodata_raster
shape = (310, 287); batch = 8; four = True
g = _nodata_raster(shape, 7 + four)
pb = int(os.environ.get("PB", "4"))
o = L.default_opts(batch=batch, precond_bytes=pb, itmax=300)
with L.raster_setup(g, o, four_neighbors=four) as h:
    info = h.info; n = info["n"]
    labels, nc = h.components()
    rng = np.random.default_rng(9)
    B = rng.standard_normal((n, 3)); B -= B.mean(axis=0)
    for c in range(nc):
        m = labels == c
        B[m] -= B[m].mean(axis=0)
    out = {"knob": os.environ.get("KNOB", ""), "pb": pb, "level_n": info["level_n"][:info["levels"]], "level_nnz": info["level_nnz"][:info["levels"]]}
    try:
        X, st = h.solve_rhs(B)
        out["rhs"] = (st["total_iters"], st["max_relres"])
    except Exception as e:
        out["rhs"] = "FAIL " + str(e)[:80]
    if os.environ.get("FINGERPRINT"):
        fp = []
        for l in range(info["levels"]):
            A = h.level_matrix(l, "A")
            rec = {"l": l, "nnz": int(A.nnz), "abs": float(abs(A).sum()), "diagmin": float(A.diagonal().min()), "diagmax": float(A.diagonal().max())}
            if l < info["levels"] - 1:
                P = h.level_matrix(l, "P")
                rec.update({"Pnnz": int(P.nnz), "Pabs": float(abs(P).sum()), "Pmin": float(P.data.min()) if P.nnz else 0.0})
            if A.shape[0] <= 2000:
                ev = np.linalg.eigvalsh(A.toarray().astype(np.float64))
                rec["ev"] = [float(ev[0]), float(ev[1]), float(ev[-1])]
            fp.append(rec)
        out["fp"] = fp
    print(json.dumps(out), flush=True)
