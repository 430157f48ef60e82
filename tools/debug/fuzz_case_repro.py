"""Re-run one case of tools/fuzz_rasters.py (sizes 60..220) with the coarse-space enrichment off and on: iterations, not_converged,\nresidual -- usage: fuzz_case_repro.py SEED CASE  (env CSGPU_LIB)"""
import os, sys, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import circuitscape_jl_amd
from circuitscape_jl_amd import lib as L, solver as ps
from oracle import refgraph as rg
L.load(os.environ.get('CSGPU_LIB'))
seed0=int(sys.argv[1]); case=int(sys.argv[2]); lo_,hi_=60,220
rng = np.random.default_rng(seed0 * 1000 + case)
R = int(rng.integers(lo_, hi_)); C = int(rng.integers(lo_, hi_))
sigma = float(rng.choice([0.5, 1.0, 2.5, 3.5]))
frac = float(rng.choice([0.0, 0.05, 0.2, 0.35]))
four = bool(rng.integers(0, 2)); avg = bool(rng.integers(0, 2)); pb = int(rng.choice([0, 4]))
g = np.exp(sigma * rng.standard_normal((R, C)))
g[rng.random((R, C)) < frac] = 0.0
if rng.random() < 0.3: g[rng.integers(0, R), :] = 0.0
if rng.random() < 0.3: g[:, rng.integers(0, C)] = 0.0
print(R,C,sigma,frac,four,avg,pb, flush=True)
nm = rg.construct_node_map(g, None)
W = rg.construct_graph(g, nm, avg, four)
ncomp, lab = sp.csgraph.connected_components(W, directed=False)
big = np.flatnonzero(lab == np.bincount(lab).argmax())
ids = rng.choice(big, size=4, replace=False)
src, dst = [int(ids[0]), int(ids[1])], [int(ids[2]), int(ids[3])]
for en in ("0","1"):
    os.environ["CSGPU_ENRICH"]=en
    try:
        with L.raster_setup(g, L.default_opts(batch=2, precond_bytes=pb, rtol=1e-10, atol=0.0), four_neighbors=four, avg_resistances=avg) as h:
            print("enrich",en,"vectors",h.info["enrich_vectors"], "precond", h.info["precond_bytes"], flush=True)
            try:
                Rr, _, volt, st = h.solve_pairs(src, dst, want_voltages=True)
                print("  pairs+volt:", st["total_iters"], st["max_iters"], st["not_converged"], st["max_relres"], st["polished_batches"], flush=True)
            except Exception as e:
                print("  EXC volt", str(e)[:160], flush=True)
            try:
                Rr2, _, _, st2 = h.solve_pairs(src, dst)
                print("  pairs     :", st2["total_iters"], st2["max_iters"], st2["not_converged"], st2["max_relres"], flush=True)
            except Exception as e:
                print("  EXC pairs", str(e)[:160], flush=True)
    except Exception as e:
        print("  EXC setup", str(e)[:200])
