#!/usr/bin/env python3
"""A network WITH locality at scale (VERDICT r3 missing #4): random geometric graph in the unit square, n nodes, mean degree
~10, node ids carry no locality, conductances U(0.5, 2) -- what hashed MIS(2) + the CSR kernels cost there: setup, levels,
iterations and ms per batch of 16 pair solves (fp64 and fp32 hierarchy), residual check.
usage: network_locality_bench.py N"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import circuitscape_jl_amd  # noqa: E402,F401
from circuitscape_jl_amd import lib as L  # noqa: E402
from test_gpu_scale import _random_graph_laplacian  # noqa: E402

L.load(os.environ.get("CSGPU_LIB"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
t0 = time.perf_counter()
G, rng = _random_graph_laplacian(n, "geo", 777)
t_gen = time.perf_counter() - t0
n = G.shape[0]
A = G.copy()
A.data = A.data + np.finfo(np.float64).eps * np.linalg.norm(A.data)      # core.jl:161
focal = rng.choice(n, size=17, replace=False)
src, dst = [int(focal[0])] * 16, [int(q) for q in focal[1:]]
for pb in (0, 4):
    t0 = time.perf_counter()
    with L.setup(A, L.default_opts(batch=16, precond_bytes=pb), index_dtype=np.int32, index_base=0) as h:
        t_setup = time.perf_counter() - t0
        info = h.info
        h.solve_pairs(src, dst)
        t1 = time.perf_counter()
        R, _, _, st = h.solve_pairs(src, dst)
        ms = (time.perf_counter() - t1) * 1e3
        print(json.dumps({"graph": "random geometric, mean degree ~10, shuffled ids", "n": int(n), "nnz": int(A.nnz),
                          "host_generation_s": t_gen, "precond_bytes": info["precond_bytes"], "levels": info["levels"],
                          "level_n": info["level_n"], "operator_complexity": info["operator_complexity"],
                          "setup_wall_s": t_setup, "setup_device_s": info["setup_ms"] / 1e3, "upload_s": info["upload_ms"] / 1e3,
                          "ms_per_batch16": ms, "iters_mean": st["total_iters"] / 16.0, "iters_max": st["max_iters"],
                          "max_relres": st["max_relres"], "not_converged": st["not_converged"], "R0": float(R[0])}), flush=True)
