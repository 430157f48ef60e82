#!/usr/bin/env python3
"""Fuzzer for the streamed set-up of host matrices (csgpu.hip, setup_from_host_streamed: what csgpu_setup does with 2^31
stored entries and more; CSGPU_STREAM_HOST_CSR sends small matrices down the same code): random rasters -- size, sigma, 0..40 %
NODATA, whole NODATA rows / columns, 4- / 8-neighbour -- largest connected component as the reference's solve() would hand
it over (compact numbering + raster cell of every node), Int64 / Int32, 1- / 0-based, random block sizes from "a handful of
rows" to "everything at once", fp64 / fp32 hierarchy. Per case: the streamed handle against the ordinary handle of the
same arrays (bit-identical resistances and iteration counts when both run the cell-space lattice form; 1e-8 otherwise) and
against a direct solve of the component (scipy). Test infrastructure only.
usage: fuzz_streamed.py SEED NCASES    (env CSGPU_LIB: library to load, default the emulator build; FUZZ_MIN / FUZZ_MAX)"""
import os, sys
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import circuitscape_jl_amd  # noqa
from circuitscape_jl_amd import lib as L, solver as ps
from oracle import refgraph as rg
L.load(os.environ.get("CSGPU_LIB", os.path.join(ROOT, "tests", "emu", "libcsgpu_emu.so")))
seed0 = int(sys.argv[1]); ncase = int(sys.argv[2])
lo_, hi_ = int(os.environ.get("FUZZ_MIN", "8")), int(os.environ.get("FUZZ_MAX", "64"))
bad = streamed = 0
for case in range(ncase):
    if case and case % 20 == 0: print("# seed", seed0, "cases done", case, "streamed", streamed, "bad", bad, flush=True)
    rng = np.random.default_rng(seed0 * 1000 + case)
    R = int(rng.integers(lo_, hi_)); C = int(rng.integers(lo_, hi_))
    sigma = float(rng.choice([0.5, 1.0, 2.5])); frac = float(rng.choice([0.0, 0.0, 0.05, 0.2, 0.4]))
    four = bool(rng.integers(0, 2)); pb = int(rng.choice([0, 4]))
    idt, base = (np.int64, 1) if rng.random() < 0.5 else (np.int32, 0)
    g = np.exp(sigma * rng.standard_normal((R, C)))
    g[rng.random((R, C)) < frac] = 0.0
    if rng.random() < 0.3: g[rng.integers(0, R), :] = 0.0
    if rng.random() < 0.3: g[:, rng.integers(0, C)] = 0.0
    if (g > 0).sum() < 40: continue
    nm = rg.construct_node_map(g, None)
    W = rg.construct_graph(g, nm, False, four)
    A = sp.csr_matrix(rg.laplacian(W))
    _, lab = sp.csgraph.connected_components(W, directed=False)
    big = np.flatnonzero(lab == np.bincount(lab).argmax())
    if len(big) < 40: continue
    Ac = sp.csr_matrix(A[big][:, big], copy=True)
    Ac.data = Ac.data + np.finfo(np.float64).eps * np.linalg.norm(Ac.data)   # the reference's shift (core.jl:161)
    row, col = ps._node_coords(nm, big + 1)
    ids = rng.choice(len(big), size=4, replace=False)
    src, dst = [int(ids[0]), int(ids[1])], [int(ids[2]), int(ids[3])]
    Rd = []
    for s, d in zip(src, dst):
        b = np.zeros(len(big)); b[d] = 1.0; b[s] = -1.0
        x = spla.spsolve(Ac.tocsc(), b); Rd.append(x[d] - x[s])
    Rd = np.array(Rd)
    block = int(rng.choice([64, 200, int(rng.integers(64, Ac.nnz + 64)), 10 ** 9]))
    tag = dict(case=case, R=R, C=C, sigma=sigma, frac=frac, four=four, pb=pb, idx=np.dtype(idt).name, base=base, block=block)
    try:
        os.environ.pop("CSGPU_STREAM_HOST_CSR", None)
        with L.setup(Ac, L.default_opts(batch=2, precond_bytes=pb, rtol=1e-10, atol=0.0), node_row=row, node_col=col,
                     index_dtype=idt, index_base=base) as h:
            Rb, _, _, stb = h.solve_pairs(src, dst); ib = h.info
        os.environ["CSGPU_STREAM_HOST_CSR"] = str(block)
        with L.setup(Ac, L.default_opts(batch=2, precond_bytes=pb, rtol=1e-10, atol=0.0), node_row=row, node_col=col,
                     index_dtype=idt, index_base=base) as h:
            Rs, _, _, sts = h.solve_pairs(src, dst); is_ = h.info
        os.environ.pop("CSGPU_STREAM_HOST_CSR", None)
        es, eb = float(np.max(np.abs(Rs - Rd) / Rd)), float(np.max(np.abs(Rb - Rd) / Rd))
        ok = es < 1e-6 and eb < 1e-6 and sts["not_converged"] == 0 and is_["n"] == len(big) and is_["nnz"] == Ac.nnz
        if is_["host_blocks"] > 0:
            streamed += 1
            same_form = ib["lattice_period"] == is_["lattice_period"] and ib["level_n"] == is_["level_n"] and len(big) < R * C
            if same_form and ib["level_n"][0] > len(big):   # the ordinary twin ran the cell-space form from the same lattice form
                ok = ok and np.array_equal(Rs, Rb) and sts["total_iters"] == stb["total_iters"]
            else:
                ok = ok and float(np.max(np.abs(Rs - Rb) / Rb)) < 1e-8
        else:
            ok = ok and np.array_equal(Rs, Rb)   # declined: the very same path
        if not ok:
            bad += 1
            print("BAD", tag, es, eb, is_["host_blocks"], ib["level_n"], is_["level_n"], stb["total_iters"], sts["total_iters"], flush=True)
    except Exception as ex:
        bad += 1
        os.environ.pop("CSGPU_STREAM_HOST_CSR", None)
        print("EXC", tag, str(ex)[:200], flush=True)
print("seed", seed0, "cases", ncase, "streamed", streamed, "bad", bad)
