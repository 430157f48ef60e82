#!/bin/bash
# NOT RUN in round 5 (the device budget was spent when these were written): the first device act for whoever has device time.
# The 22 device tests added after the round's last device run -- mgVerify7, the streamed set-up of host matrices, the 20
# pairwise goldens in precision = single -- then the streamed set-up at a size where it matters: a 6000 x 6000 raster with
# 10 % NODATA handed over as Int64 / 1-based CSR arrays with node coordinates (3.2e8 stored entries, 5.2 GB of host arrays),
# ordinary path against CSGPU_STREAM_HOST_CSR=2^26 (5 blocks): resistances, iteration counts, set-up wall time.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5p
mkdir -p $OUT
timeout 600 python -m pytest tests/test_zz_streamed_host_csr.py "tests/test_gpu_golden.py::test_raster_advanced_on_gpu" -m gpu -q \
    > $OUT/pytest_new_device_tests.log 2>&1; tail -5 $OUT/pytest_new_device_tests.log
FUZZ_MIN=60 FUZZ_MAX=220 CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu.so timeout 600 python tools/fuzz_streamed.py 7 60 \
    > $OUT/fuzz_streamed_device.log 2>&1; tail -2 $OUT/fuzz_streamed_device.log
timeout 900 python - > $OUT/streamed_6000.json 2> $OUT/streamed_6000.err <<'PY'
import json, os, sys, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import circuitscape_jl_amd  # noqa
from circuitscape_jl_amd import lib as L
L.load(os.environ.get("CSGPU_LIB")) if os.environ.get("CSGPU_LIB") else L.load()
R = C = int(os.environ.get("STREAM_TEST_SIZE", "6000"))
rng = np.random.default_rng(5)
g = np.exp(rng.standard_normal((R, C))); g[rng.random((R, C)) < 0.10] = 0.0
h = L.raster_setup(g, L.default_opts(batch=8, precond_bytes=4))          # graph, components and numbering on the device
lab, nc = h.components(); A = h.level_matrix(0, "A"); nm = np.asarray(h.raster_nodemap()); h.close()
big = np.flatnonzero(lab == np.bincount(lab).argmax())
Ac = sp.csr_matrix(A)[big][:, big]
rr, cc = np.nonzero(nm); ids_ = nm[rr, cc] - 1
row_of = np.empty(len(lab), dtype=np.int32); col_of = np.empty(len(lab), dtype=np.int32); row_of[ids_] = rr; col_of[ids_] = cc
row, col = row_of[big], col_of[big]
ids = rng.choice(len(big), 16, replace=False); src = [int(v) for v in ids[:8]]; dst = [int(v) for v in ids[8:]]
out = {"n": int(len(big)), "nnz": int(Ac.nnz)}
for name, env in (("ordinary", None), ("streamed", str(1 << 26))):
    os.environ.pop("CSGPU_STREAM_HOST_CSR", None)
    if env: os.environ["CSGPU_STREAM_HOST_CSR"] = env
    t = time.perf_counter()
    with L.setup(Ac, L.default_opts(batch=8, precond_bytes=4), node_row=row, node_col=col) as hh:
        wall = time.perf_counter() - t; i = hh.info; Rr, _, _, st = hh.solve_pairs(src, dst)
    out[name] = {"setup_wall_s": wall, "upload_ms": i["upload_ms"], "setup_ms": i["setup_ms"], "host_blocks": i["host_blocks"],
                 "level_n": i["level_n"], "iters": st["total_iters"], "R": Rr.tolist()}
out["max_rel_diff"] = float(np.max(np.abs(np.array(out["ordinary"]["R"]) - np.array(out["streamed"]["R"])) / np.array(out["ordinary"]["R"])))
print(json.dumps(out))
PY
tail -c 600 $OUT/streamed_6000.json; tail -3 $OUT/streamed_6000.err
