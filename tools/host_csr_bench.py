"""PCIe-inclusive setup through csgpu_setup (host CSR, Int64 / 1-based like Julia) vs the in-HBM raster build."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import circuitscape_jl_amd  # noqa
from circuitscape_jl_amd import lib
from oracle import refgraph as rg, refsolve as rs
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
g = 1.0 / np.exp(np.random.default_rng(12345).standard_normal((size, size)))
t = time.time(); A = rs.regularize(rg.raster_laplacian_from_conductance(g)); t_host_build = time.time() - t
rows = (np.arange(size * size) % size).astype(np.int32); cols = (np.arange(size * size) // size).astype(np.int32)
t = time.time(); h = lib.setup(A, lib.default_opts(precond_bytes=4), node_row=rows, node_col=cols); t_setup = time.time() - t
i = h.info
cells = np.random.default_rng(67890).choice(size * size, size=9, replace=False)
R, _, _, st = h.solve_pairs([cells[0]] * 8, list(cells[1:]))
h2 = lib.raster_setup(g, lib.default_opts(precond_bytes=4)); R2, _, _, st2 = h2.solve_pairs([cells[0]] * 8, list(cells[1:]))
print(json.dumps({"size": size, "nnz": int(A.nnz), "host_scipy_build_s": t_host_build, "csgpu_setup_wall_s": t_setup,
                  "upload_ms": i["upload_ms"], "upload_bytes": int(A.nnz * 16 + (A.shape[0] + 1) * 8 + 8 * A.shape[0]),
                  "amg_setup_ms": i["setup_ms"], "raster_setup_upload_ms": h2.info["upload_ms"],
                  "max_abs_R_diff": float(np.max(np.abs(R - R2))), "iters": st["max_iters"]}))
