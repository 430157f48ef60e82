#!/usr/bin/env python3
"""BASELINE configs[4] network (n = 5e6, 5e7 edges): polynomial degree of the single-level preconditioner x batch width.
One graph, one process; per (batch, sweeps): set-up, one warm-up batch, 3 timed batches of one-to-all sources through
csgpu_solve_sources (check voltages + cumulative current vector). Prints one JSON line per combination."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import circuitscape_jl_amd  # noqa: F401,E402
from circuitscape_jl_amd import lib  # noqa: E402


def main():
    import torch
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000000
    lib.load(os.environ.get("CSGPU_LIB"))
    dev = torch.device("cuda", 0) if torch.cuda.is_available() else None
    G, rng = bench.random_network(n, torch=torch if dev is not None else None, dev=dev)
    n = G.shape[0]
    ref = None
    for K in (16, 32):
        for sweeps in (8, 4, 2, 1, -1):
            focal = np.random.default_rng(7).choice(n, size=4 * K, replace=False).reshape(4, K)
            src, gnd, chk = [], [], []
            for b in range(4):
                s_, g_, c_ = bench.one_to_all_columns(focal[b])
                src += s_
                gnd += g_
                chk += c_
            t0 = time.perf_counter()
            h = lib.setup(G, lib.default_opts(batch=K, precond_bytes=4, itmax=5000, last_level_sweeps=sweeps),
                          index_dtype=np.int32, index_base=0)
            t_setup = time.perf_counter() - t0
            info = h.info
            h.solve_sources(src[:K], gnd[:K], check=chk[:K])
            cum = np.zeros(n)
            t0 = time.perf_counter()
            v, _, _, st = h.solve_sources(src[K:], gnd[K:], check=chk[K:], cum=cum)
            wall = time.perf_counter() - t0
            h.close()
            if ref is None:
                ref = {}
            key = K
            ref.setdefault(key, v)
            calls = max(st["cg_spmv_calls"], 1)
            print(json.dumps({"batch": st["batch"], "sweeps_opt": sweeps, "sweeps_effective": info["last_level_sweeps"],
                              "levels": info["levels"], "sources": 3 * K, "iters_mean": st["total_iters"] / (3.0 * K),
                              "iters_max": st["max_iters"], "wall_s": wall, "device_s": st["device_ms"] / 1e3,
                              "sources_per_s_device": 3 * K / (st["device_ms"] / 1e3), "sources_per_s_wall": 3 * K / wall,
                              "setup_s": t_setup, "setup_device_s": info["setup_ms"] / 1e3, "max_relres": st["max_relres"],
                              "not_converged": st["not_converged"], "cg_spmm_ms": st["cg_spmv_ms"] / calls,
                              "cg_spmm_GBs": st["cg_spmv_bytes"] / (st["cg_spmv_ms"] / calls * 1e-3) / 1e9,
                              "max_rel_diff_vs_first": float(np.max(np.abs(v - ref[key]) / np.abs(ref[key])))}), flush=True)


if __name__ == "__main__":
    main()
