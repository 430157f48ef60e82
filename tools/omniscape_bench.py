"""Moving-window throughput (scope row N3): W circular windows of diameter D solved (a) one by one, each with its own
graph build + AMG setup + solve (what a loop over compute_omniscape_current does), (b) stacked into one raster and
solved as ONE block-diagonal system. Usage (GPU box): python tools/omniscape_bench.py [D] [W]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import circuitscape_jl_amd  # noqa: F401,E402
from circuitscape_jl_amd import lib, solver as ps  # noqa: E402
from test_emu_solver import _omniscape_window  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 201
W = int(sys.argv[2]) if len(sys.argv) > 2 else 64
wins = [_omniscape_window(D, 1000 + k) for k in range(W)]
cfg = {"connect_four_neighbors_only": "False"}
s = ps.HIPAMGSolver(bs=1, opts={"precond_bytes": 4})
ps.compute_omniscape_current_batch(wins[:2], cfg, solver=s)        # warm-up
t0 = time.perf_counter()
seq = [ps.compute_omniscape_current_batch([w], cfg, solver=s)[0][0] for w in wins]
t1 = time.perf_counter()
bat, st = ps.compute_omniscape_current_batch(wins, cfg, solver=s)
t2 = time.perf_counter()
err = max(float(np.max(np.abs(a - b)) / max(1e-30, np.max(np.abs(a)))) for a, b in zip(seq, bat))
print(json.dumps({"diameter": D, "windows": W, "cells_per_window": int((wins[0][0] > 0).sum()),
                  "one_by_one_s": t1 - t0, "windows_per_s_one_by_one": W / (t1 - t0), "batched_s": t2 - t1,
                  "windows_per_s_batched": W / (t2 - t1), "batched_iters": st["max_iters"],
                  "batched_polished": st["polished_batches"], "max_rel_diff_batched_vs_single": err}))
