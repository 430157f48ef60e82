#!/usr/bin/env python3
"""A REAL host matrix with 2^31 stored entries and more through csgpu_setup (the reference's
use_64bit_indexing, src/run.jl:34): the 8-neighbour Laplacian of an all-valid 16000 x 16000 raster (2.56e8 nodes, 2.30e9
stored entries: Int64 colptr / rowval, Float64 nzval, 1-based -- 39 GB of host arrays, built here with numpy the way
construct_graph / laplacian! build it, src/raster/pairwise.jl:316-362, src/core.jl:608-634, regularised like core.jl:161),
handed over with the raster cell of every node (csgpu_opts.node_row / node_col). The library streams it in blocks of 2^28
entries into the lattice form (csgpu.hip, setup_from_host_streamed). Resistances of 4 pairs against csgpu_raster_setup of the
same raster. Log: profiles/r6_host_csr_2e31.json."""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def host_laplacian(g):
    """(rowptr, colidx, vals) Int64 / Int64 / Float64, 1-based, rows sorted: the reference's graph of an all-valid raster
    in column-major node numbering (node = j * R + i), average conductance, diagonal edges / sqrt(2); every stored entry
    shifted by eps(Float64) * norm(nzval)."""
    R, C = g.shape
    n = R * C
    P = np.zeros((C + 2, R + 2))
    P[1:-1, 1:-1] = g.T                                   # P[j + 1, i + 1] = g[i, j]
    ctr = P[1:-1, 1:-1]
    vals = np.zeros((n, 9))
    cols = np.empty((n, 9), dtype=np.int64)
    keep = np.zeros((n, 9), dtype=bool)
    node = np.arange(n, dtype=np.int64)
    k = 0
    for dj in (-1, 0, 1):
        for di in (-1, 0, 1):
            if dj == 0 and di == 0:
                cols[:, k] = node + 1
                keep[:, k] = True
            else:
                nb = P[1 + dj:C + 1 + dj, 1 + di:R + 1 + di]
                w = (ctr + nb) * (0.5 / np.sqrt(2.0) if (dj != 0 and di != 0) else 0.5)
                ok = nb > 0
                vals[:, k] = np.where(ok, -w, 0.0).ravel()
                keep[:, k] = ok.ravel()
                cols[:, k] = node + (dj * R + di) + 1
            k += 1
    vals[:, 4] = -vals.sum(axis=1)
    counts = keep.sum(axis=1, dtype=np.int64)
    rowptr = np.empty(n + 1, dtype=np.int64)
    rowptr[0] = 1
    np.cumsum(counts, out=rowptr[1:])
    rowptr[1:] += 1
    colidx = cols[keep]
    nz = vals[keep]
    del cols, vals, keep
    nz += np.finfo(np.float64).eps * np.sqrt(np.dot(nz, nz))
    return rowptr, colidx, nz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=16000)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r6_host_csr_2e31.json"))
    args = ap.parse_args()
    import bench
    import circuitscape_jl_amd  # noqa: F401
    from circuitscape_jl_amd import lib
    L = lib.load(os.environ.get("CSGPU_LIB"))
    N = args.size
    g = bench.make_raster(N)
    t0 = time.time()
    rp, ci, va = host_laplacian(g)
    t_build = time.time() - t0
    n, nnz = N * N, int(len(ci))
    out = {"size": N, "n": n, "nnz": nnz, "nnz_ge_2_31": bool(nnz >= 2 ** 31), "host_build_s": t_build,
           "host_bytes": int(rp.nbytes + ci.nbytes + va.nbytes), "index_type": "Int64, 1-based", "value_type": "Float64"}
    print(json.dumps(out), flush=True)
    cells, pairs = bench.focal_pairs(N)
    src = [p[0] for p in pairs[:4]]
    dst = [p[1] for p in pairs[:4]]
    node = np.arange(n, dtype=np.int64)
    nrow = np.ascontiguousarray(node % N, dtype=np.int32)
    ncol = np.ascontiguousarray(node // N, dtype=np.int32)
    del node
    o = lib.default_opts(batch=4)
    o.node_row = nrow.ctypes.data
    o.node_col = ncol.ctypes.data
    hp = ctypes.c_void_p(0)
    t0 = time.time()
    rc = L.csgpu_setup(rp.ctypes.data, ci.ctypes.data, va.ctypes.data, n, nnz, 8, 8, 1, ctypes.byref(o), ctypes.byref(hp))
    t_setup = time.time() - t0
    if rc != 0:
        out["failed"] = (L.csgpu_last_error() or b"").decode()
        json.dump(out, open(args.out, "w"), indent=1)
        print(json.dumps(out))
        return
    h = lib.Handle(hp, np.float64)
    info = h.info
    R1, _, _, st1 = h.solve_pairs(src, dst)
    h.close()
    del rp, ci, va
    out["streamed"] = {"setup_wall_s": t_setup, "upload_s": info["upload_ms"] / 1e3, "device_setup_s": info["setup_ms"] / 1e3,
                       "host_blocks": info["host_blocks"], "lattice_period": info["lattice_period"], "levels": info["levels"],
                       "device_bytes": info["device_bytes"], "R": [float(x) for x in R1],
                       "iters_mean": st1["total_iters"] / 4.0, "max_relres": st1["max_relres"], "not_converged": st1["not_converged"]}
    t0 = time.time()
    h2 = lib.raster_setup(g, lib.default_opts(batch=4))
    t2 = time.time() - t0
    R2, _, _, st2 = h2.solve_pairs(src, dst)
    i2 = h2.info
    h2.close()
    rel = float(np.max(np.abs(R1 - R2) / np.abs(R2)))
    out["raster_entry_point"] = {"setup_wall_s": t2, "device_setup_s": i2["setup_ms"] / 1e3, "R": [float(x) for x in R2],
                                 "iters_mean": st2["total_iters"] / 4.0, "levels": i2["levels"]}
    out.update({"max_rel_diff_R": rel, "tolerance": 1e-8, "ok": bool(rel < 1e-8 and info["host_blocks"] > 1)})
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
