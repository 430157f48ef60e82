#!/usr/bin/env python3
"""The CPU baseline MEASURED at the full size: the oracle (C++ restatement of the reference CG+AMG path,
oracle/cs_oracle.cpp) on BASELINE configs[2]'s own raster -- 10000 x 10000, bench.make_raster, regularised like core.jl:161 --
at the reference's tolerances (Krylov rule, rtol 1e-6), one pair per host thread as the reference parallelises
(src/core.jl:262-272), THREADS pairs (default 16: a pair's work vectors are ~6 GB). Writes the JSON bench.py cites as
cpu_baseline.measured_full_size (profiles/r5_cpu_baseline_full_size.json). Needs ~150 GB of host memory and a few minutes;
no GPU work.  usage: cpu_full_size.py [SIZE] [OUT.json]   env: THREADS=16"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import refgraph as rg, refsolve as rs  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r5_cpu_baseline_full_size.json")
T = int(os.environ.get("THREADS", "16"))
t0 = time.time()
G = rg.raster_laplacian_from_conductance(bench.make_raster(N))
A = rs.regularize(G)
del G
t_graph = time.time() - t0
t0 = time.time()
S = rs.OracleAMG(A)
t_setup = time.time() - t0
cells, pairs = bench.focal_pairs(N)
src, dst = [p[0] for p in pairs[:T]], [p[1] for p in pairs[:T]]
t0 = time.time()
R1, _, res1 = S.solve_pairs(src[:1], dst[:1])
t_one = time.time() - t0
t0 = time.time()
R, _, res = S.solve_pairs(src, dst, nthreads=T)
t_mt = time.time() - t0
per_pair = t_mt / T
out = {"what": "oracle (CPU restatement of the reference CG+AMG path) at the full size of BASELINE configs[2], reference tolerances",
       "size": N, "n": int(A.shape[0]), "nnz": int(A.nnz), "host_cores": os.cpu_count(), "threads": T,
       "graph_build_s": t_graph, "setup_s": t_setup, "one_pair_one_thread_s": t_one, "iters_one": res1[0]["iters"],
       "pairs": T, "pairs_wall_s": t_mt, "iters": [r["iters"] for r in res],
       "value_pair_solves_per_s": 1.0 / (per_pair + t_setup / 100.0),
       "single_thread_value": 1.0 / (t_one + t_setup / 100.0),
       "R": [float(v) for v in R]}
with open(out_path, "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "R"}))
