"""GPU micro-benchmark of the fine-level SpMV/SpMM kernel variants (tuning aid, not part of the product)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import circuitscape_jl_amd  # noqa
from circuitscape_jl_amd import lib
if os.environ.get('CSGPU_LIB'):
    lib.load(os.environ['CSGPU_LIB'])
size = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
prec = sys.argv[2] if len(sys.argv) > 2 else "double"
dt = np.float64 if prec == "double" else np.float32
g = (1.0 / np.exp(np.random.default_rng(12345).standard_normal((size, size)))).astype(dt)
h = lib.raster_setup(g, lib.default_opts(max_levels=1))
info = h.info
vb = 8 if dt == np.float64 else 4
out = {"variant": os.environ.get("CSGPU_SPMM_VARIANT", "default"), "size": size, "prec": prec}
ks = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [1, 2, 4, 8, 16]
for k in ks:
    ms = h.spmv_bench(k, 10)
    b = info["nnz"] * (vb + 4) + (info["n"] + 1) * 4 + 2 * info["n"] * k * vb
    out["k%d" % k] = {"ms": round(ms, 4), "GBs": round(b / ms / 1e6, 1)}
print(json.dumps(out))
