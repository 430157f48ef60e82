#!/usr/bin/env python3
"""Where does the wall time of a second handle in one process go? (bench.py's fp64 leg saw a 3.9 s setup after the
mixed handle had been closed.) Times setup / first batch / second batch / close for two handles in sequence."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import circuitscape_jl_amd  # noqa
from circuitscape_jl_amd import lib
lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
g = bench.make_raster(N)
cells, pairs = bench.focal_pairs(N)
src = [p[0] for p in pairs[:16]]; dst = [p[1] for p in pairs[:16]]
out = []
for name, pb in (("mixed", 4), ("fp64", 0), ("mixed_again", 4)):
    t0 = time.perf_counter(); h = lib.raster_setup(g, lib.default_opts(batch=16, precond_bytes=pb)); t1 = time.perf_counter()
    i = h.info
    h.solve_pairs(src, dst); t2 = time.perf_counter()
    h.solve_pairs(src, dst); t3 = time.perf_counter()
    h.close(); t4 = time.perf_counter()
    out.append(dict(name=name, setup_wall=t1 - t0, upload_ms=i["upload_ms"], setup_ms=i["setup_ms"], first_batch=t2 - t1,
                    second_batch=t3 - t2, close=t4 - t3, device_gb=i["device_bytes"] / 1e9))
    print(json.dumps(out[-1]), flush=True)
