#!/usr/bin/env python3
"""Fuzzer for the enriched levels on the fused residual update + restriction (round 6; csrc/enrich.h enrich_coarse_setup /
enrich_coarse_fix): random rasters -- 30..260 cells a side (every residue mod 3: last tiles of 2, 3 and 4 cells), 5..35 %
NODATA i.i.d. plus whole NODATA rows / columns and walls, log-normal sigma 0.3..1.5, enrichment threshold 0.06..0.2, batches
of 16 / 32 with a ragged tail -- solved with the fused pass (csgpu_opts.fused_restrict = 1) and with two passes (-1) on handles
that differ in nothing else: same per-pair iteration counts (+-1 on at most one pair in fifty), resistances equal to 1e-7,
both equal to a direct solve of the regularised system to 1e-5, the fused handle's second set-up gives the same bits.
usage: fuzz_enrich_fused.py SEED NCASES    (env CSGPU_LIB: library to load, default the emulator build)"""
import json, os, sys
import numpy as np, scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import circuitscape_jl_amd  # noqa: E402,F401
from circuitscape_jl_amd import lib as L  # noqa: E402
from oracle import refgraph as rg, refsolve as rs  # noqa: E402  (input generation and checking only)
L.load(os.environ.get("CSGPU_LIB", os.path.join(ROOT, "tests", "emu", "libcsgpu_emu.so")))
seed0, ncase = int(sys.argv[1]), int(sys.argv[2])
lo_, hi_ = int(os.environ.get("FUZZ_MIN", "30")), int(os.environ.get("FUZZ_MAX", "260"))
bad = 0
for case in range(ncase):
    rng = np.random.default_rng(seed0 * 1000 + case)
    R, C = int(rng.integers(lo_, hi_)), int(rng.integers(lo_, hi_))
    sigma = float(rng.choice([0.3, 0.7, 1.0, 1.5]))
    frac = float(rng.uniform(0.05, 0.35))
    g = np.exp(sigma * rng.standard_normal((R, C)))
    g[rng.random((R, C)) < frac] = 0.0
    for _ in range(int(rng.integers(0, 4))):   # walls with a gap
        if rng.random() < 0.5:
            r = int(rng.integers(1, R - 1)); g[r, :] = 0.0; g[r, int(rng.integers(0, C))] = 1.0
        else:
            c = int(rng.integers(1, C - 1)); g[:, c] = 0.0; g[int(rng.integers(0, R)), c] = 1.0
    tau = float(rng.choice([0.06, 0.1, 0.15, 0.2]))
    batch = int(rng.choice([16, 32]))
    npairs = int(rng.integers(16, 45))
    rec = {"case": case, "shape": [R, C], "sigma": sigma, "frac": round(frac, 3), "tau": tau, "batch": batch, "pairs": npairs}
    try:
        out = {}
        src = dst = None
        for tag, fused in (("two", -1), ("fused", 1), ("fused2", 1)):
            with L.raster_setup(g, L.default_opts(batch=batch, precond_bytes=0, enrich=0, enrich_tau=tau, fused_restrict=fused,
                                                  stream=-1)) as h:
                info = h.info
                if src is None:
                    labels, _ = h.components()
                    big = np.flatnonzero(labels == np.bincount(labels).argmax())
                    if big.size < 2 * npairs + 2:
                        break
                    ids = rng.choice(big, size=2 * npairs, replace=False)
                    src, dst = [int(v) for v in ids[:npairs]], [int(v) for v in ids[npairs:]]
                Rr, _, _, st = h.solve_pairs(src, dst)
                out[tag] = (Rr, st["total_iters"], st["not_converged"], h.info["fused_restrict_solves"], info["enrich_vectors"],
                            info["level_form"][0])
        if src is None:
            rec["skipped"] = "giant component too small"
            print(json.dumps(rec), flush=True)
            continue
        two, fa, fb = out["two"], out["fused"], out["fused2"]
        rec.update(enrich_vectors=fa[4], level0_form=fa[5], fused_batches=fa[3], iters_two=two[1], iters_fused=fa[1])
        ok = two[2] == 0 and fa[2] == 0 and two[3] == 0
        ok = ok and np.array_equal(fa[0], fb[0]) and fa[1] == fb[1]
        rel = float(np.max(np.abs(fa[0] - two[0]) / np.abs(two[0])))
        rec["rel_fused_vs_two"] = rel
        ok = ok and rel < 1e-7 and abs(fa[1] - two[1]) <= max(1, npairs // 50)
        # direct solve of the reference's regularised system, three pairs
        nm = rg.construct_node_map(g, None)
        A = rs.regularize(rg.laplacian(rg.construct_graph(g, nm, False, False))).tocsr()
        pos = -np.ones(A.shape[0], dtype=np.int64); pos[big] = np.arange(big.size)   # (the giant component's block: the others
        lu = spla.splu(A[big][:, big].tocsc())                                        #  may be single nodes whose pivot is the shift)
        worst = 0.0
        for k in range(3):
            b = np.zeros(big.size); b[pos[dst[k]]] = 1.0; b[pos[src[k]]] = -1.0
            x = lu.solve(b)
            Rd = x[pos[dst[k]]] - x[pos[src[k]]]
            worst = max(worst, abs(fa[0][k] - Rd) / abs(Rd))
        rec["rel_vs_direct"] = worst
        ok = ok and worst < 1e-5
        rec["ok"] = bool(ok)
    except Exception as e:  # noqa: BLE001
        rec["ok"] = False
        rec["error"] = repr(e)[:300]
    bad += 0 if rec.get("ok", True) else 1
    print(json.dumps(rec), flush=True)
print(json.dumps({"seed": seed0, "cases": ncase, "bad": bad}))
