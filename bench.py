#!/usr/bin/env python3
"""bench.py -- pair-solves/sec + CG-SpMV GB/s vs the HBM roofline on the synthetic raster pairwise problem.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches
it under torch.distributed.run with one rank per GPU. One JSON line is printed by rank 0.

Workload (BASELINE.json configs[2]): R x R all-valid raster (default 10000 x 10000), 8-neighbour,
average conductance, resistances r = exp(N(0,1)) (seed 12345), g = 1/r; the Laplacian is built directly in HBM
(csgpu_raster_setup) and regularised like the reference (core.jl:161); 15 focal cells (seed 67890) -> the
lexicographic pair list (105 pairs, the first 100 are the config's "100 focal pairs").

A step = one batch of `--batch` (default 16) pair solves (AMG-preconditioned CG, the reference's stopping rule: rtol 1e-6 /
atol sqrt(eps) on sqrt(r'M^-1 r), core.jl:639) through csgpu_solve_pairs. AMG setup happens once per matrix (as in
the reference, core.jl:164) before the timed region; its cost is reported separately AND amortised into `value`
over the config's 100 pairs:  value = pairs / (t_steps + t_setup * pairs/100).

N > 1 (weak scaling): every rank holds the full matrix + hierarchy (the path shards by independent pairs,
SURVEY.md 8e), solves its own K batches, and the resistances are gathered with one RCCL all_gather.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


KERNEL_SOURCES = ("stencil.h", "lattice.h", "spmv.h", "blas1.h", "prims.h", "common.h", "pcg.h", "dia25.h", "amg_setup.h",
                  "lattice_setup.h", "enrich.h")


def kernel_source_hash():
    """sha256[:16] over the csrc files that define the roofline kernels (the marching lattice kernels and the CSR SpMM).
    A committed PMC pass is only reported as `roofline.traffic` while it describes THIS code: tools/pmc_update.py stores the
    hash next to the counters, and a mismatch prints null (the figure used to be keyed by workload
    only and went stale silently after a kernel change)."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "circuitscape.jl_amd", "csrc", f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(size, batch, vb, lattice=False, pb=4):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (tools/gpu_pmc.sh + tools/pmc_update.py
    -> profiles/pmc_traffic.json), collected with rocprofv3 --pmc in separate passes and corrected as MI355X_MICROARCH.md
    prescribes. Returns (bytes or None, note): None when no profile matches the current workload OR the entry was taken
    with different kernel sources (entry["kernel_src_sha16"] != kernel_source_hash()). (PMC passes cannot run inside this
    process; the key names size, batch width, CG precision, matrix form and the preconditioner precision.)"""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    key = "%d_k%d_f%d%s%s" % (size, batch, vb * 8, "_lattice" if lattice else "",
                              "_fp64precond" if (vb == 8 and pb == 8) else "")
    try:
        with open(path) as f:
            e = json.load(f).get(key)
        if not e or e.get("traffic_bytes_per_launch") is None:
            return None, "no committed PMC pass for %s" % key
        cur = kernel_source_hash()
        if e.get("kernel_src_sha16") != cur:
            return None, ("committed PMC pass %s was taken with kernel sources %s, this build is %s: stale, not reported"
                          % (key, e.get("kernel_src_sha16"), cur))
        return e["traffic_bytes_per_launch"], "PMC pass %s (kernel sources %s)" % (key, cur)
    except Exception as ex:
        return None, "pmc_traffic.json unreadable: %r" % (ex,)


def live_pmc_traffic(args, precond, kernel_prefix, timeout_s=170):
    """one kernel: see live_pmc_traffic_many"""
    return live_pmc_traffic_many(args, precond, [kernel_prefix], timeout_s)[kernel_prefix]


def live_pmc_traffic_many(args, precond, kernel_prefixes, timeout_s=170):
    """`roofline.traffic` measured in THIS run: two rocprofv3 passes -- `--pmc FETCH_SIZE` and `--pmc
    WRITE_SIZE`, separately, with `--kernel-trace` only, as MI355X_MICROARCH.md's HBM section prescribes -- over a child
    process that runs one warm-up and one timed batch of the same workload on the same path, nothing else. HBM bytes per
    launch of the roofline kernel = FETCH_SIZE x 2 (the counter's unit is 64 B on gfx950 while rocprofv3 scales it as 32 B;
    verified on streaming kernels of known size, profiles/pmc_traffic.json) + WRITE_SIZE, both KiB, averaged over the
    full-size launches (the warm-up problem launches the same kernel on a 768^2 raster). Returns {prefix: (bytes or None,
    note)} for every kernel name prefix asked for, out of the SAME two passes."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {k: (None, "rocprofv3 not found") for k in kernel_prefixes}
    vals = {k: {} for k in kernel_prefixes}
    fail = {}
    t0 = time.perf_counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="csgpu_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
               os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--cpu-sample", "0", "--compare-steps", "0",
               "--extra-legs", "0", "--host-csr", "0", "--pmc-live", "0", "--size", str(args.size), "--batch", str(args.batch),
               "--precond", precond, "--precision", args.precision, "--calls", args.calls]
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout_s)
            got = {k: [] for k in kernel_prefixes}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") != counter:
                            continue
                        for k in kernel_prefixes:
                            if row.get("Kernel_Name", "").startswith(k):
                                got[k].append(float(row["Counter_Value"]))
            for k in kernel_prefixes:
                if not got[k]:
                    fail.setdefault(k, "live PMC pass %s: no launches of %s in the counter file" % (counter, k))
                    continue
                full = [v for v in got[k] if v >= 0.5 * max(got[k])]
                vals[k][counter] = (sum(full) / len(full), len(full))
        except Exception as e:
            for k in kernel_prefixes:
                fail.setdefault(k, "live PMC pass %s failed: %r" % (counter, e))
        finally:
            shutil.rmtree(d, ignore_errors=True)
    res = {}
    for k in kernel_prefixes:
        if k in fail or len(vals[k]) < 2:
            res[k] = (None, fail.get(k, "live PMC pass incomplete"))
            continue
        traffic = (2.0 * vals[k]["FETCH_SIZE"][0] + vals[k]["WRITE_SIZE"][0]) * 1024.0
        res[k] = (traffic, "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace "
                  "only) over a one-batch child run; FETCH x2 (gfx950 unit) + WRITE, KiB; mean over %d / %d full-size launches "
                  "of %s; %.0f s" % (vals[k]["FETCH_SIZE"][1], vals[k]["WRITE_SIZE"][1], k, time.perf_counter() - t0))
    return res


def make_raster(size, seed=12345, sigma=1.0, dtype=np.float64):
    rng = np.random.default_rng(seed)
    r = np.exp(sigma * rng.standard_normal((size, size)))
    return (1.0 / r).astype(dtype)


def focal_pairs(size, npts=15, seed=67890):
    rng = np.random.default_rng(seed)
    cells = rng.choice(size * size, size=npts, replace=False)
    pairs = [(int(cells[i]), int(cells[j])) for i in range(npts) for j in range(i + 1, npts)]
    return cells, pairs


def cpu_baseline(sample_size, nsolve=2, max_threads=32, ntight=16, single=False):
    """Oracle (CPU restatement of the reference CG+AMG path) on a bounded sample of the same workload: one thread, and
    all host cores the way the reference parallelises (one pair per task, src/core.jl:262-272). Also returns the
    TIGHT oracle's resistances (true-residual rtol 1e-12) of the first `ntight` pairs: the parity reference the GPU
    path is compared with on the same sample raster (`parity` in the bench line)."""
    from oracle import refgraph as rg, refsolve as rs
    g = make_raster(sample_size)
    G = rg.raster_laplacian_from_conductance(g)
    # (single precision: the reference's shift eps(Float32) * norm(nzval) of EVERY stored entry, core.jl:161, is part of
    # the problem definition -- at n = 1e8 it is 4e-3 per entry and turns the Laplacian into a strongly grounded system;
    # the oracle solves that same matrix, in double)
    A = rs.regularize(G, dtype=np.float32).astype(np.float64) if single else rs.regularize(G)
    t0 = time.time()
    S = rs.OracleAMG(A)
    t_setup = time.time() - t0
    cells, pairs = focal_pairs(sample_size)
    src = [p[0] for p in pairs[:nsolve]]
    dst = [p[1] for p in pairs[:nsolve]]
    t0 = time.time()
    R, _, res = S.solve_pairs(src, dst)
    t_solve = (time.time() - t0) / nsolve
    out = dict(setup_s=t_setup, solve_s=t_solve, iters=[r["iters"] for r in res], spmv_s=S.spmv_seconds(3),
               n=sample_size * sample_size, nnz=int(A.nnz), R=R.tolist())
    nthreads = max(1, min(os.cpu_count() or 1, max_threads))
    if nthreads > 1:
        npar = min(nthreads, len(pairs))
        t0 = time.time()
        _, _, res_mt = S.solve_pairs([p[0] for p in pairs[:npar]], [p[1] for p in pairs[:npar]], nthreads=npar)
        out.update(mt_threads=npar, mt_wall_s=time.time() - t0, mt_pairs=npar, mt_iters=[r["iters"] for r in res_mt])
    if ntight > 0:
        nt = min(ntight, len(pairs))
        t0 = time.time()
        Rt, _, res_t = S.solve_pairs([p[0] for p in pairs[:nt]], [p[1] for p in pairs[:nt]], rtol=1e-12, atol=0.0,
                                     criterion=1, nthreads=max(1, min(nthreads, nt)))
        out.update(tight_R=Rt.tolist(), tight_pairs=nt, tight_wall_s=time.time() - t0,
                   tight_max_true_relres=max(r["true_relres"] for r in res_t))
    return out


def cpu_full_size_measure(size, threads, single=False):
    """CPU child: the oracle (C++ restatement of the reference CG+AMG path) AT THE FULL SIZE of the workload -- the
    reference's graph construction restated (oracle/refgraph.py), regularised like core.jl:161, set-up once, then `threads`
    pairs at the reference's tolerances, one pair per host thread as the reference parallelises (src/core.jl:262-272).
    10000^2: ~20 s graph + ~47 s set-up + ~115 s for 16 pairs on 16 threads, ~150 GB of host memory."""
    from oracle import refgraph as rg, refsolve as rs
    t0 = time.time()
    G = rg.raster_laplacian_from_conductance(make_raster(size))
    A = rs.regularize(G, dtype=np.float32).astype(np.float64) if single else rs.regularize(G)
    del G
    t_graph = time.time() - t0
    t0 = time.time()
    S = rs.OracleAMG(A)
    t_setup = time.time() - t0
    _, pairs = focal_pairs(size)
    T = max(1, min(threads, len(pairs), os.cpu_count() or 1))
    t0 = time.time()
    R, _, res = S.solve_pairs([p[0] for p in pairs[:T]], [p[1] for p in pairs[:T]], nthreads=T)
    t_mt = time.time() - t0
    return {"size": size, "n": int(A.shape[0]), "nnz": int(A.nnz), "host_cores": os.cpu_count(), "threads": T,
            "graph_build_s": t_graph, "setup_s": t_setup, "pairs": T, "pairs_wall_s": t_mt, "iters": [r["iters"] for r in res],
            "value_pair_solves_per_s": 1.0 / (t_mt / T + t_setup / 100.0)}


def host_can_measure_full_size(size):
    """enough cores and memory for cpu_full_size_measure without getting in the GPU legs' way"""
    try:
        avail = 0.0
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                avail = int(line.split()[1]) / 1e6
        return size >= 8000 and (os.cpu_count() or 1) >= 32 and avail >= 200.0
    except Exception:
        return False


NODATA_SEED, NODATA_PTS_SEED = 2468, 97531


def nodata_raster(g, frac=0.15):
    """the bench raster with `frac` of its cells NODATA (i.i.d., rng(2468)): construct_node_map drops cells with
    conductance <= 0 (src/raster/pairwise.jl:271-301)"""
    rng = np.random.default_rng(NODATA_SEED)
    return np.where(rng.random(g.shape) < frac, g.dtype.type(0), g)


def lexicographic_pairs(pts):
    return [(int(pts[i]), int(pts[j])) for i in range(len(pts)) for j in range(i + 1, len(pts))]


def oracle_leg_nodata(sample_size, npairs=8, frac=0.15):
    """CHECKER side of the `nodata15` leg's parity figure (runs in the CPU child): the reference's own graph construction
    restated (oracle/refgraph.py: construct_node_map / construct_graph / laplacian, src/raster/pairwise.jl:271-362,
    src/core.jl:608-634) on a sample raster of the same generator and the same NODATA mask generator, regularised like
    core.jl:161, giant component by scipy, 15 focal nodes of it (rng(97531)) -> the first `npairs` lexicographic pairs
    solved by the TIGHT oracle. Independent of the device: the parent solves the same node pairs on the GPU."""
    import scipy.sparse.csgraph as csg
    from oracle import refgraph as rg, refsolve as rs
    t0 = time.time()
    g = nodata_raster(make_raster(sample_size), frac)
    nm = rg.construct_node_map(g, None)
    A = rs.regularize(rg.laplacian(rg.construct_graph(g, nm, False, False)))
    ncomp, lab = csg.connected_components(A, directed=False)
    giant = np.flatnonzero(lab == np.bincount(lab).argmax())
    pts = np.random.default_rng(NODATA_PTS_SEED).choice(giant, size=15, replace=False)
    pairs = lexicographic_pairs(pts)[:npairs]
    S = rs.OracleAMG(A)
    nth = max(1, min(os.cpu_count() or 1, npairs))
    R, _, res = S.solve_pairs([p[0] for p in pairs], [p[1] for p in pairs], rtol=1e-12, atol=0.0, criterion=1, nthreads=nth)
    return {"sample_size": sample_size, "n": int(A.shape[0]), "components": int(ncomp), "giant": int(giant.size),
            "src": [p[0] for p in pairs], "dst": [p[1] for p in pairs], "R": R.tolist(),
            "max_true_relres": max(r["true_relres"] for r in res), "iters": [r["iters"] for r in res],
            "wall_s": time.time() - t0}


def oracle_leg_fp32(sample_size, npairs=8):
    """CHECKER side of the `config3_fp32` leg's parity figure (CPU child): the reference's single-precision problem --
    the Float32 Laplacian with every stored entry shifted by eps(Float32) * norm(nzval) (core.jl:161), built on the host by
    the oracle's graph code -- solved in double by the TIGHT oracle for the first `npairs` bench pairs of the sample."""
    from oracle import refgraph as rg, refsolve as rs
    t0 = time.time()
    G = rg.raster_laplacian_from_conductance(make_raster(sample_size, dtype=np.float32).astype(np.float64))
    A = rs.regularize(G, dtype=np.float32).astype(np.float64)
    _, pairs = focal_pairs(sample_size)
    pairs = pairs[:npairs]
    S = rs.OracleAMG(A)
    nth = max(1, min(os.cpu_count() or 1, npairs))
    R, _, res = S.solve_pairs([p[0] for p in pairs], [p[1] for p in pairs], rtol=1e-12, atol=0.0, criterion=1, nthreads=nth)
    return {"sample_size": sample_size, "n": int(A.shape[0]), "src": [p[0] for p in pairs], "dst": [p[1] for p in pairs],
            "R": R.tolist(), "max_true_relres": max(r["true_relres"] for r in res), "wall_s": time.time() - t0}


def cpu_baseline_entry(cb, n_full, size, sample_size):
    """The `cpu_baseline` object of the bench line from a cpu_baseline() measurement."""
    scale = float(n_full) / cb["n"]
    cpu_value = 1.0 / (cb["solve_s"] * scale + cb["setup_s"] * scale / 100.0)
    sample = ("oracle (C++ restatement of the reference CG+AMG path) on a %dx%d raster of the same generator: setup "
              "%.2fs + %d pair solves at %.2fs on one thread (%s iterations); time scaled linearly in n (x%.0f) to "
              "the %dx%d workload, setup amortised over 100 pairs"
              % (sample_size, sample_size, cb["setup_s"], len(cb["iters"]), cb["solve_s"], cb["iters"], scale, size, size))
    entry = {"value": cpu_value, "unit": "pair-solves/s", "cores": 1, "kind": "port", "sample": sample,
             "spmv_GBs": (cb["nnz"] * 12 + (cb["n"] + 1) * 4 + 2 * cb["n"] * 8) / cb["spmv_s"] / 1e9}
    if "mt_threads" in cb:
        # all host cores, one pair per thread as the reference does (core.jl:262-272); the setup stays serial
        per_pair = cb["mt_wall_s"] / cb["mt_pairs"]
        entry["single_thread_value"] = cpu_value
        entry["value"] = 1.0 / (per_pair * scale + cb["setup_s"] * scale / 100.0)
        entry["cores"] = cb["mt_threads"]
        entry["sample"] = sample + ("; then %d pairs on %d threads (one pair per thread) in %.2fs"
                                    % (cb["mt_pairs"], cb["mt_threads"], cb["mt_wall_s"]))
    entry["sample_n"] = cb["n"]
    # the same oracle MEASURED at the full size on a GPU box's host cores (tools/cpu_full_size.py; minutes of CPU work, so
    # not part of the default run): committed with its provenance, cited here beside the live bounded-sample figure
    try:
        with open(os.path.join(ROOT, "profiles", "r5_cpu_baseline_full_size.json")) as f:
            m = json.load(f)
        if m.get("size") == size:
            entry["measured_full_size"] = {
                "value": m["value_pair_solves_per_s"], "unit": "pair-solves/s", "cores": m["threads"], "kind": "port, measured at full size",
                "setup_s": m["setup_s"], "pairs": m["pairs"], "pairs_wall_s": m["pairs_wall_s"], "iters": m["iters"],
                "single_thread_value": m["single_thread_value"], "host_cores": m["host_cores"],
                "source": "profiles/r5_cpu_baseline_full_size.json (tools/cpu_full_size.py on a GPU box of this round)"}
    except Exception:
        pass
    if "tight_R" in cb:  # consumed by the parent (parity), not part of the published object
        entry["_tight"] = {"R": cb["tight_R"], "pairs": cb["tight_pairs"], "max_true_relres": cb["tight_max_true_relres"],
                           "wall_s": cb["tight_wall_s"]}
    return entry


def run_pairs(h, batch_pairs, K, Wm, sync=None, first_batch=0, one_call=True):
    """Wm untimed + K timed steps (a step = one batch of pairs) through csgpu_solve_pairs; returns (elapsed_s, resistances
    per step, stats sums). one_call: the pairs of all K timed steps are handed over in ONE call -- what a host does with
    a component's pair list (solve(prob, ::HIPAMGSolver, ...) sends the whole list down) -- so the library can stream them
    through its columns (pcg_stream_pairs: a column takes the next pair as soon as its own has converged); otherwise
    one call per step (every batch runs at the pace of its slowest column)."""
    def pairs_of(steps):
        s, d = [], []
        for k in steps:
            a, b = batch_pairs(k)
            s += a
            d += b
        return s, d
    if Wm > 0:
        if one_call:
            h.solve_pairs(*pairs_of(range(first_batch, first_batch + Wm)))
        else:
            for w in range(Wm):
                h.solve_pairs(*batch_pairs(first_batch + w))
    if sync:
        sync()
    t0 = time.perf_counter()
    results = []
    agg = dict(total_iters=0, max_iters=0, cg_spmv_ms=0.0, cg_spmv_calls=0, device_ms=0.0, max_relres=0.0,
               not_converged=0, stream_slots=0, calls=0, resid_ms=0.0, resid_calls=0, resid_bytes=0, resid_fused=0)
    calls = [range(first_batch + Wm, first_batch + Wm + K)] if one_call else [[first_batch + Wm + k] for k in range(K)]
    for steps in calls:
        s, d = pairs_of(steps)
        R, _, _, st = h.solve_pairs(s, d)
        per = len(R) // len(steps)
        results += [R[i * per:(i + 1) * per] for i in range(len(steps))]
        agg["total_iters"] += st["total_iters"]
        agg["max_iters"] = max(agg["max_iters"], st["max_iters"])
        agg["cg_spmv_ms"] += st["cg_spmv_ms"]
        agg["cg_spmv_calls"] += st["cg_spmv_calls"]
        agg["device_ms"] += st["device_ms"]
        agg["max_relres"] = max(agg["max_relres"], st["max_relres"])
        agg["not_converged"] += st["not_converged"]
        agg["cg_spmv_bytes"] = st["cg_spmv_bytes"]
        agg["resid_ms"] += st.get("resid_ms", 0.0)
        agg["resid_calls"] += st.get("resid_calls", 0)
        agg["resid_bytes"] = st.get("resid_bytes", 0)
        agg["resid_fused"] = st.get("resid_fused", 0)
        agg["stream_slots"] += st.get("stream_slots", 0)
        agg["calls"] += 1
    return time.perf_counter() - t0, results, agg


def stream_block(agg, npairs, B):
    """How the timed pairs went through the library's columns: K-wide iterations of the stream and the columns'
    utilisation (a pair costs its own iterations + 1 slots: the +1 is its initial V-cycle), or the batch path's waste."""
    if agg.get("stream_slots", 0) > 0:
        return {"mode": "streaming (pcg_stream_pairs: a column takes the next pair when its own has converged)",
                "slots": agg["stream_slots"], "column_utilisation": (agg["total_iters"] + npairs) / float(agg["stream_slots"] * B)}
    return {"mode": "batches (every batch runs until its slowest column has converged)", "slots": 0,
            "iters_mean_over_max": agg["total_iters"] / float(max(npairs, 1)) / max(agg["max_iters"], 1)}


def cg_product_name(info, B, vb):
    """Name of the fine-level CG product kernel (the roofline kernel); its algorithmic bytes per launch are reported by
    the library (csgpu_stats.cg_spmv_bytes, formulas in include/csgpu.h and DESIGN.md section 4)."""
    tn = {8: "double", 4: "float"}
    xb = info["precond_bytes"] or vb
    if info.get("lattice_period", 0) > 0:
        return "dia_cg_kernel<%s,%s,%d,CG> (lattice-form CG product, p-update fused)" % (tn[vb], tn[xb], B)
    return "spmv_kernel<%s,%d,PLAIN,DOT,x=%s> (fine-level CSR CG SpMM)" % (tn[vb], B, tn[xb])


def config3_fp32_leg(lib, size, B, dev_index, batch_pairs, sync, one_call):
    """BASELINE configs[3] precision on one GPU: the all-fp32 handle (val_bytes = 4, the reference's `precision = single`,
    src/run.jl:29) with the LIBRARY DEFAULTS = the reference's (regularisation eps(Float32) * norm(nzval) of every stored
    entry, rtol 1e-6, atol sqrt(eps(Float32)), the 1e-4 check) on the bench raster; 1 warm-up + 3 timed steps. Parity of
    this path at 5000^2 against the tight oracle on the same fp32-shifted matrix: tests/test_gpu_scale.py (2.3e-7)."""
    g32 = make_raster(size, dtype=np.float32)
    h = lib.raster_setup(g32, lib.default_opts(device=dev_index, batch=B))
    try:
        el, res, agg = run_pairs(h, batch_pairs, 3, 1, sync, one_call=one_call)
        info = h.info
    finally:
        h.close()
    sv = (info["setup_ms"] + info["upload_ms"]) / 1e3
    return {"value": 3 * B / (el + sv * 3 * B / 100.0), "unit": "pair-solves/s", "dtype": "f32", "steps": 3,
            "ms_per_16_pairs": el / 3 * 1e3 * 16.0 / B, "iters_mean": agg["total_iters"] / float(3 * B),
            "max_relres": agg["max_relres"], "not_converged": agg["not_converged"], "setup_s": sv,
            "note": "the reference's single-precision problem: every stored entry shifted by eps(Float32) * norm(nzval) "
                    "(core.jl:161), ~4e-3 per entry at n = 1e8 -- a strongly grounded system, hence the low iteration count"}


def nodata_leg(lib, g, B, make_opts, precond, sync, frac=0.15, steps=2):
    """The bench raster with 15 % of its cells NODATA (i.i.d.; construct_node_map drops cells with conductance <= 0,
    src/raster/pairwise.jl:271-301) -- what a real landscape looks like next to the all-valid headline: cell-space handle
    (the full lattice on the marching kernels, NODATA cells as weightless rows), 3x3 tiles refined by the piece analysis,
    coarse levels in 25-point lattice form (csrc/dia25.h). Focal cells: 15 cells of the giant component; 1 warm-up batch +
    `steps` timed batches of B pairs in one call, on the `value` path's precision. Parity of this path against the tight
    oracle: tests/test_gpu_parity.py (2000^2)."""
    gh = nodata_raster(g, frac)
    h = lib.raster_setup(gh, make_opts(precond))
    try:
        del gh
        info = h.info
        labels, ncomp = h.components()
        giant = np.flatnonzero(labels == np.bincount(labels).argmax())
        pts = np.random.default_rng(NODATA_PTS_SEED).choice(giant, size=15, replace=False)
        pairs = lexicographic_pairs(pts)

        def batch_pairs(k):
            idx = [(k * B + i) % len(pairs) for i in range(B)]
            return [pairs[i][0] for i in idx], [pairs[i][1] for i in idx]
        el, res, agg = run_pairs(h, batch_pairs, steps, 1, sync, one_call=True)
        fused = h.info.get("fused_restrict_solves", 0)
    finally:
        h.close()
    sv = (info["setup_ms"] + info["upload_ms"]) / 1e3
    return {"value": steps * B / (el + sv * steps * B / 100.0), "unit": "pair-solves/s", "nodata_fraction": frac,
            "fused_restrict_batches": int(fused),   # > 0: the enriched level took the fused residual update + restriction (enrich.h)
            "nodes": int(info["n"]), "giant_component_nodes": int(giant.size), "components": int(ncomp),
            "lattice_period": info["lattice_period"], "levels": info["levels"], "level_form": info["level_form"],
            "enrich_vectors": info.get("enrich_vectors", 0), "steps": steps,
            "ms_per_16_pairs": el / steps * 1e3 * 16.0 / B, "iters_mean": agg["total_iters"] / float(steps * B),
            "iters_max": agg["max_iters"], "max_relres": agg["max_relres"], "not_converged": agg["not_converged"],
            "setup_s": sv, "precond": precond}


def leg_parity(lib, g, opts, tight, tolerance, what):
    """GPU side of a leg's parity figure: the leg's own options on the CPU child's sample raster, the child's node pairs,
    against the TIGHT oracle's resistances (a parity figure accompanies every number)."""
    h = lib.raster_setup(g, opts)
    try:
        info = h.info
        if info["n"] != tight["n"]:
            return {"failed": "node count differs: device %d, oracle graph %d" % (info["n"], tight["n"])}
        R, _, _, st = h.solve_pairs(tight["src"], tight["dst"])
    finally:
        h.close()
    Ro = np.asarray(tight["R"])
    rel = float(np.max(np.abs(np.asarray(R, dtype=np.float64) - Ro) / np.abs(Ro)))
    return {"max_rel_err": rel, "tolerance": tolerance, "ok": bool(rel < tolerance), "pairs": len(Ro), "n": int(info["n"]),
            "sample": what, "oracle": "tight (true-residual rtol 1e-12) on the host-built graph of the same sample",
            "oracle_max_true_relres": tight["max_true_relres"], "iters_mean": st["total_iters"] / float(len(Ro)),
            "not_converged": int(st["not_converged"]), "lattice_period": info["lattice_period"],
            "level_form": info["level_form"]}


def _csr_laplacian_from_edges(lo, hi, w, n, torch=None, dev=None):
    """CSR Laplacian (sorted rows, diagonal included) of the undirected weighted graph {lo[k], hi[k]}: w[k]. The 2 m + n
    (key, value) entries are sorted once by key = row * n + col -- on the GPU through torch when one is visible (input
    GENERATION only, outside every timed region: sorting 1.05e8 keys takes the host ~15 s at BASELINE's 5e6 nodes and the
    device 0.1 s; both routes give the identical matrix), else by numpy."""
    import scipy.sparse as sp
    deg = np.bincount(lo, weights=w, minlength=n) + np.bincount(hi, weights=w, minlength=n)
    ar = np.arange(n, dtype=np.int64)
    key = np.concatenate([lo * n + hi, hi * n + lo, ar * n + ar])
    val = np.concatenate([-w, -w, deg])
    if torch is not None and dev is not None:
        k = torch.from_numpy(key).to(dev)
        k, order = torch.sort(k)
        v = torch.from_numpy(val).to(dev)[order]
        rows = torch.div(k, n, rounding_mode="floor")
        cols = (k - rows * n).to(torch.int32).cpu().numpy()
        counts = torch.bincount(rows, minlength=n)
        indptr = np.zeros(n + 1, dtype=np.int64)
        indptr[1:] = torch.cumsum(counts, 0).cpu().numpy()
        val = v.cpu().numpy()
        del k, order, v, rows, counts
    else:
        order = np.argsort(key, kind="stable")
        key = key[order]
        val = val[order]
        rows = key // n
        cols = (key - rows * n).astype(np.int32)
        indptr = np.zeros(n + 1, dtype=np.int64)
        indptr[1:] = np.cumsum(np.bincount(rows, minlength=n))
    G = sp.csr_matrix((val, cols, indptr.astype(np.int32 if indptr[-1] < 2**31 else np.int64)), shape=(n, n))
    G.has_sorted_indices = True
    return G


def random_network(n, seed=424242, torch=None, dev=None):
    """BASELINE configs[4] generator (tools/network_bench.py, tests/test_gpu_scale.py): 10 n endpoint pairs from rng(seed),
    deduplicated, conductances U(0.5, 2); the giant component is kept (mean degree 20: the graph is connected with
    probability 1 - 1e-2 at n = 5e6; checked, and cut down to the giant component if it is not)."""
    import scipy.sparse.csgraph as csg
    rng = np.random.default_rng(seed)
    i = rng.integers(0, n, size=10 * n)
    j = rng.integers(0, n, size=10 * n)
    keep = i != j
    lo, hi = np.minimum(i[keep], j[keep]), np.maximum(i[keep], j[keep])
    del i, j, keep
    key = lo.astype(np.int64) * n + hi
    if torch is not None and dev is not None:
        key = torch.unique(torch.from_numpy(key).to(dev), sorted=True).cpu().numpy()
    else:
        key = np.unique(key)
    lo, hi = key // n, key % n
    del key
    w = rng.uniform(0.5, 2.0, size=len(lo))
    G = _csr_laplacian_from_edges(lo, hi, w, n, torch, dev)
    ncomp, lab = csg.connected_components(G, directed=False)
    if ncomp > 1:
        giant = np.flatnonzero(lab == np.bincount(lab).argmax())
        G = G[giant][:, giant].tocsr()
        G.sort_indices()
    return G, rng


def geometric_network(n, seed=777, torch=None, dev=None):
    """A network WITH locality (tests/test_gpu_scale.py): random geometric graph in the unit square, mean degree ~10, node
    ids carry no locality, conductances U(0.5, 2); giant component."""
    import scipy.sparse.csgraph as csg
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(seed)
    pts = rng.random((n, 2))
    pr = cKDTree(pts).query_pairs(np.sqrt(10.0 / (np.pi * n)), output_type="ndarray")
    lo, hi = np.minimum(pr[:, 0], pr[:, 1]).astype(np.int64), np.maximum(pr[:, 0], pr[:, 1]).astype(np.int64)
    w = rng.uniform(0.5, 2.0, size=len(lo))
    G = _csr_laplacian_from_edges(lo, hi, w, n, torch, dev)
    ncomp, lab = csg.connected_components(G, directed=False)
    giant = np.flatnonzero(lab == np.bincount(lab).argmax())
    G = G[giant][:, giant].tocsr()
    G.sort_indices()
    return G, rng


def _masked_jacobi_cg(G, b, ground, rtol=1e-12, maxiter=2000):
    """Independent host check of one one-to-all column: scipy's CG with a Jacobi preconditioner on the REDUCED system
    (rows / columns of the grounded nodes removed, src/raster/advanced.jl:282-288), expressed as a masked operator so that
    the 1e8-entry matrix is not sliced; true residual driven to rtol."""
    import scipy.sparse.linalg as spla
    n = G.shape[0]
    free = np.ones(n)
    free[ground] = 0.0
    dinv = free / G.diagonal()
    op = spla.LinearOperator((n, n), matvec=lambda v: free * (G @ (free * v)), dtype=np.float64)
    x, flag = spla.cg(op, free * b, rtol=rtol, atol=0.0, maxiter=maxiter, M=spla.LinearOperator((n, n), matvec=lambda v: dinv * v, dtype=np.float64))
    res = float(np.linalg.norm(free * (G @ x) - free * b) / np.linalg.norm(free * b))
    return x, flag, res


def csr_spmm_roofline(st, info, K):
    """roofline object of the CSR SpMM that is the CG product of a handle without lattice structure (networks): algorithmic
    bytes per launch nnz (val + 4) + (n + 1) 4 + n K (x + val) (reported by the library) / the mean HIP-event
    duration of those launches in this call."""
    calls = max(st["cg_spmv_calls"], 1)
    avg_ms = st["cg_spmv_ms"] / calls
    nbytes = st["cg_spmv_bytes"]
    ach = nbytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    tv = "float" if info["val_bytes"] == 4 else "double"
    tx = "float" if (info["precond_bytes"] or info["val_bytes"]) == 4 else "double"
    return {"bound": "hbm", "kernel": "csgpu::spmv_kernel<%s, %d, PLAIN, DOT, x = %s> (CSR SpMM: the CG product A p of a network; "
                                      "matrix values %s, gathered search direction %s)" % (tv, K, tx, tv, tx),
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
            "traffic_source": "not collected in this run (profiles/r6_network_* hold the rocprofv3 --pmc passes)",
            "algorithmic_bytes_per_launch": nbytes, "avg_ms": avg_ms, "launches_timed": st["cg_spmv_calls"]}


def one_to_all_columns(focal):
    """advanced one-to-all over the focal nodes (src/raster/onetoall.jl:106-117): column s = unit current into focal[s],
    every other focal node tied to ground; the check node of a column is its source (`res[i] = v[1]`, onetoall.jl:141)."""
    focal = [int(q) for q in focal]
    return [[p] for p in focal], [[q for q in focal if q != p] for p in focal], focal


def config4_network_leg(lib, dev_index, n=5000000, nsrc=32, ncheck=2, torch=None, dev=None, batch=32):
    """BASELINE configs[4] at its stated size on one GPU (5e6 nodes, 5e7 undirected edges -> 1.05e8 stored entries):
    network mode, advanced one-to-all (src/raster/advanced.jl:274-312, src/network/advanced.jl:1-51) -- unit current at one
    focal node, the other focal nodes tied to ground, every source a column of ONE csgpu_solve_sources call on ONE handle:
    sparse right-hand sides in (a column is a single +1), the sources' voltages and the cumulative node-current vector out
    (what the one-to-all driver keeps, onetoall.jl:141,153-158) -- no n x nrhs array crosses PCIe in the timed call. Reported
    wall AND device (HIP-event time of the PCG loops) with a roofline of the CSR SpMM.
    This random graph is an expander: the setup declines to coarsen it, so the preconditioner is JACOBI (levels = 1: one
    damped-Jacobi sweep after the scaling -- a degree-1 polynomial, csgpu_opts.last_level_sweeps), not AMG; batches of 32
    columns (an fp32 hierarchy gathers one full 128-byte line of x per stored entry at that width).
    The first `ncheck` columns are solved again with their voltages handed back (untimed) and compared with an
    independent scipy Jacobi-CG solve of the reduced system at true-residual 1e-12 (parity: resistance to the grounded set +
    whole voltage vector), and their true residual is evaluated on the host."""
    t0 = time.perf_counter()
    G, rng = random_network(n, torch=torch, dev=dev)
    t_gen = time.perf_counter() - t0
    n = G.shape[0]
    focal = rng.choice(n, size=nsrc, replace=False)
    src, grounds, chk = one_to_all_columns(focal)
    t0 = time.perf_counter()
    h = lib.setup(G, lib.default_opts(device=dev_index, batch=batch, precond_bytes=4, itmax=2000), index_dtype=np.int32,
                  index_base=0)
    t_setup = time.perf_counter() - t0
    try:
        info = h.info
        h.solve_sources(src[:1], grounds[:1], check=chk[:1])          # (untimed: code objects, work vectors)
        cum = np.zeros(n)
        t0 = time.perf_counter()
        v, _, _, st = h.solve_sources(src, grounds, check=chk, cum=cum)
        t_solve = time.perf_counter() - t0
        nck = min(ncheck, nsrc)
        vck, X, _, stck = h.solve_sources(src[:nck], grounds[:nck], check=chk[:nck], want_voltages=True)
        K = st["batch"]
    finally:
        h.close()
    t0 = time.perf_counter()
    rerr, verr, cres, worst = 0.0, 0.0, 0.0, 0.0
    for s_ in range(nck):
        b = np.zeros(n)
        b[focal[s_]] = 1.0
        r = G @ X[:, s_] - b
        r[grounds[s_]] = 0.0
        worst = max(worst, float(np.linalg.norm(r)))
        xs, flag, res = _masked_jacobi_cg(G, b, grounds[s_])
        cres = max(cres, res if flag == 0 else float("inf"))
        rerr = max(rerr, abs(X[focal[s_], s_] - xs[focal[s_]]) / abs(xs[focal[s_]]),
                   abs(v[s_] - xs[focal[s_]]) / abs(xs[focal[s_]]))   # (the timed call's answer for the same column too)
        verr = max(verr, float(np.max(np.abs(X[:, s_] - xs)) / np.max(np.abs(xs))))
    parity = {"max_rel_err": float(max(rerr, verr)), "max_rel_err_resistance": float(rerr), "max_rel_err_voltages": float(verr),
              "tolerance": 1e-6, "ok": bool(max(rerr, verr) < 1e-6), "columns_checked": nck,
              "checker": "scipy CG + Jacobi on the reduced system (masked operator), true residual %.1e" % cres,
              "check_s": time.perf_counter() - t0}
    iters = st["total_iters"] / float(nsrc)
    nbatches = -(-nsrc // max(K, 1))
    dev_s = st["device_ms"] / 1e3
    return {"value": nsrc / (t_setup + t_solve), "unit": "one-to-all sources/s (setup + solves, wall)",
            "value_device": nsrc / (info["setup_ms"] / 1e3 + dev_s),
            "value_device_note": "sources / (device time of the set-up + HIP-event time of the PCG loops)",
            "n": int(n), "nnz": int(G.nnz),
            "undirected_edges": int((G.nnz - n) // 2), "sources": nsrc, "batch": K, "levels": info["levels"],
            "preconditioner": "AMG" if info["levels"] > 1 else "Jacobi polynomial, %d sweep(s) (expander: the setup declines to "
                              "coarsen, amg_setup.h)" % info["last_level_sweeps"],
            "setup_s": t_setup, "setup_device_s": info["setup_ms"] / 1e3, "upload_s": info["upload_ms"] / 1e3,
            "solve_s_all_sources": t_solve, "solve_device_s_all_sources": dev_s,
            "pcg_device_ms_per_iteration": st["device_ms"] / max(st["max_iters"] * nbatches, 1),
            "boundary": "csgpu_solve_sources: sparse right-hand sides in; %d source voltages + one cumulative node-current "
                        "n-vector (%.0f MB) out" % (nsrc, n * 8 / 1e6),
            "iters_mean": iters, "iters_max": st["max_iters"], "max_relres_device": st["max_relres"],
            "not_converged": st["not_converged"], "worst_true_residual_norm": worst, "cum_current_sum": float(cum.sum()),
            "generate_s": t_gen, "parity": parity, "roofline": csr_spmm_roofline(st, info, K)}


def network_geometric_leg(lib, dev_index, tight, n=1000000, torch=None, dev=None):
    """A network that really coarsens, under the driver: random geometric graph (n = 1e6, mean degree 10, shuffled ids),
    hashed MIS(2) aggregation + CSR kernels (no raster coordinates), pair solves from one anchor as network pairwise mode
    does (src/network/pairwise.jl:31-65) against the TIGHT oracle's resistances computed by the CPU child on the same
    graph (same generator, host route)."""
    from oracle import refsolve as rs
    G, rng = geometric_network(n, torch=torch, dev=dev)
    A = rs.regularize(G)
    if A.shape[0] != tight["n"]:
        return {"failed": "node count differs: %d vs oracle graph %d" % (A.shape[0], tight["n"])}
    src, dst = tight["src"], tight["dst"]
    t0 = time.perf_counter()
    h = lib.setup(A, lib.default_opts(device=dev_index, batch=8, precond_bytes=0), index_dtype=np.int32, index_base=0)
    t_setup = time.perf_counter() - t0
    try:
        info = h.info
        h.solve_pairs(src, dst)
        t0 = time.perf_counter()
        R, _, _, st = h.solve_pairs(src, dst)
        t_solve = time.perf_counter() - t0
    finally:
        h.close()
    Ro = np.asarray(tight["R"])
    rel = float(np.max(np.abs(R - Ro) / np.abs(Ro)))
    return {"value": len(src) / (t_setup + t_solve), "unit": "pair-solves/s (setup + solves)", "n": int(A.shape[0]), "nnz": int(A.nnz),
            "levels": info["levels"], "level_n": info["level_n"], "operator_complexity": info["operator_complexity"],
            "preconditioner": "AMG (hashed MIS(2) aggregation, CSR kernels)", "setup_s": t_setup, "setup_device_s": info["setup_ms"] / 1e3,
            "solve_s": t_solve, "pairs": len(src), "iters_mean": st["total_iters"] / float(len(src)), "max_relres": st["max_relres"],
            "not_converged": st["not_converged"], "solve_device_s": st["device_ms"] / 1e3,
            "value_device": len(src) / ((info["setup_ms"] + st["device_ms"]) / 1e3), "batch": st["batch"],
            "roofline": csr_spmm_roofline(st, info, st["batch"]),
            "parity": {"max_rel_err": rel, "tolerance": 1e-6, "ok": bool(rel < 1e-6), "pairs": len(src),
                       "oracle": "tight (true-residual rtol 1e-12) on the same graph", "oracle_max_true_relres": tight["max_true_relres"]}}


def oracle_leg_geometric(n=1000000, npairs=8):
    """CHECKER side of `network_geometric` (CPU child): the same generator through the host route, regularised like
    core.jl:161, 8 pairs from one anchor solved by the TIGHT oracle."""
    from oracle import refsolve as rs
    t0 = time.time()
    G, rng = geometric_network(n)
    A = rs.regularize(G)
    focal = rng.choice(A.shape[0], size=npairs + 1, replace=False)
    src = [int(focal[0])] * npairs
    dst = [int(q) for q in focal[1:]]
    S = rs.OracleAMG(A)
    R, _, res = S.solve_pairs(src, dst, rtol=1e-12, atol=0.0, criterion=1, nthreads=max(1, min(os.cpu_count() or 1, npairs)))
    return {"n": int(A.shape[0]), "src": src, "dst": dst, "R": R.tolist(), "max_true_relres": max(r["true_relres"] for r in res),
            "iters": [r["iters"] for r in res], "wall_s": time.time() - t0}


def rank_identity(torch, rank, dev_index, has_cuda, ms_per_step, pairs_done, agg):
    """Per-rank facts for the N-GPU line: which physical device this rank ran on and what it did."""
    import socket
    bus, name = None, None
    if has_cuda:
        try:
            pr = torch.cuda.get_device_properties(dev_index)
            name = pr.name
            if hasattr(pr, "pci_bus_id"):
                bus = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, getattr(pr, "pci_device_id", 0))
        except Exception:
            pass
        if bus is None:
            try:
                import ctypes
                hip = ctypes.CDLL("libamdhip64.so")
                buf = ctypes.create_string_buffer(64)
                if hip.hipDeviceGetPCIBusId(buf, 64, int(dev_index)) == 0:
                    bus = buf.value.decode()
            except Exception:
                pass
    return {"rank": int(rank), "host": socket.gethostname(), "device_ordinal": int(dev_index), "pci_bus_id": bus,
            "device_name": name, "ms_per_step": float(ms_per_step), "pairs_done": int(pairs_done),
            "iters_mean": agg["total_iters"] / float(max(pairs_done, 1)), "max_relres": float(agg["max_relres"]),
            "not_converged": int(agg["not_converged"])}


def shortcut_leg(h, cells, setup_s, check_pairs):
    """what a real no-map run does (src/core.jl:137-146, 563-587, 685-739) -- K - 1 solves from the anchor
    point with the K focal voltages gathered on the device, then R_ij = R_1i + R_1j - 2 v_i^(1j) (update_shortcut_resistances!)
    for all K (K - 1) / 2 pairs. One untimed + one timed pass; a sample of the derived resistances is checked against direct
    solves of the same pairs on the same handle."""
    anchor = int(cells[0])
    others = [int(c) for c in cells[1:]]
    gather = [int(c) for c in cells]
    h.solve_pairs([anchor] * len(others), others, gather=gather)
    t0 = time.perf_counter()
    R1, Gv, _, st = h.solve_pairs([anchor] * len(others), others, gather=gather)
    t = time.perf_counter() - t0
    npts = len(cells)
    R = np.zeros((npts, npts))
    for j in range(1, npts):
        R[0, j] = R[j, 0] = R1[j - 1]
    for j in range(1, npts):           # solve p = j - 1 has its sink at point j; Gv[p, i] = v_i - v_anchor
        for i in range(1, j):
            R[i, j] = R[j, i] = R1[i - 1] + R1[j - 1] - 2.0 * Gv[j - 1, i]
    src = [p[0] for p in check_pairs]
    dst = [p[1] for p in check_pairs]
    Rd, _, _, _ = h.solve_pairs(src, dst)
    idx = {int(c): k for k, c in enumerate(cells)}
    rel = max(abs(R[idx[a], idx[b]] - rd) / abs(rd) for a, b, rd in zip(src, dst, Rd))
    nres = npts * (npts - 1) // 2
    return {"value": nres / (t + setup_s), "unit": "pair resistances/s, whole shortcut job (setup + K-1 anchor solves)",
            "focal_points": npts, "solves": len(others), "resistances": nres, "solve_s": t, "setup_s": setup_s,
            "iters_mean": st["total_iters"] / float(len(others)), "max_relres": st["max_relres"],
            "max_rel_diff_vs_direct_solves": float(rel), "direct_pairs_checked": len(src)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=7)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=10000)
    ap.add_argument("--batch", type=int, default=32,
                    help="columns solved together per pass = pairs per step (32 since round 4: the matrix values of every "
                         "marching pass are amortised over twice as many columns, +8 %% fp64 / +10 %% mixed pair-solves/s "
                         "over 16 on one box, profiles/r4_batch16_vs_32_*.json)")
    ap.add_argument("--calls", default="single", choices=["single", "per-step"],
                    help="single (default): the pairs of all timed steps go down in ONE csgpu_solve_pairs call, as a host "
                         "hands over a component's pair list, and the library streams them through its columns; per-step: "
                         "one call per batch (every batch at the pace of its slowest column)")
    ap.add_argument("--precision", default="double", choices=["double", "single"])
    ap.add_argument("--cpu-sample", type=int, default=3000, help="raster edge of the CPU-baseline / parity sample (0 = skip)")
    ap.add_argument("--criterion", type=int, default=0)
    ap.add_argument("--precond", default="same", choices=["same", "fp32"],
                    help="precision of the AMG preconditioner (hierarchy, V-cycle, stored search direction) of the path "
                         "`value` is measured on: the same precision as the CG iteration (default: the reference computes "
                         "everything in T = Float64, src/run.jl:29, src/core.jl:639), or fp32 under the fp64 CG iteration. "
                         "The other one is timed beside it over --compare-steps steps")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo only to exercise the "
                         "multi-rank code path on a box with fewer GPUs than ranks)")
    ap.add_argument("--opt", action="append", default=[], help="extra csgpu_opts override key=value (tuning)")
    ap.add_argument("--compare-steps", type=int, default=-1,
                    help="N=1 only: also time this many steps on the OTHER preconditioner precision (see --precond) and "
                         "report them as value_mixed / value_fp64 with their own roofline; -1 = the same number as "
                         "--steps, 0 = skip")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default, what the driver runs): every rank solves --steps batches. strong: the config's "
                         "FIXED pair list (--pairs, default 100 = BASELINE configs[2]) is dealt over the ranks at pair "
                         "granularity; --steps is ignored and the line reports the whole job incl. the per-rank setup")
    ap.add_argument("--pairs", type=int, default=100, help="--scaling strong: total number of pairs of the job")
    ap.add_argument("--host-csr", type=int, default=1,
                    help="N=1 only: also time csgpu_setup from host CSR arrays the way Julia hands them (Int64, 1-based) "
                         "at the bench size and report setup_host_csr_s (needs ~30 GB of host memory and ~20 s at "
                         "10000^2; 0 = skip)")
    ap.add_argument("--extra-legs", type=int, default=1,
                    help="N=1 only: also report value_shortcut (K-1 anchor solves + focal-voltage gather -> all K(K-1)/2 "
                         "resistances, SURVEY.md 8d) and value_with_voltages (whole solution vector carried, explicit 1e-4 "
                         "check); 0 = skip")
    ap.add_argument("--cpu-baseline-only", type=int, default=0, metavar="N_FULL",
                    help="internal: run only the CPU-baseline leg on a --cpu-sample raster, scale to N_FULL nodes, print "
                         "its JSON object and exit (the bench runs this in a child process so that nothing on the host "
                         "side can take the GPU line down)")
    ap.add_argument("--pmc-live", type=int, default=1,
                    help="N=1 only: measure roofline.traffic in this run with two rocprofv3 --pmc passes over a one-batch child "
                         "process (~1 min); 0 = report the committed PMC pass of profiles/pmc_traffic.json when its kernel-source "
                         "hash matches this build")
    ap.add_argument("--cpu-legs", default="", help="internal (CPU child): comma list of nodata,fp32,geometric")
    ap.add_argument("--leg-sample", type=int, default=2000,
                    help="raster edge of the samples on which the nodata15 / config3_fp32 legs are compared with the tight oracle")
    ap.add_argument("--workload", default="raster", choices=["raster", "network"],
                    help="raster = BASELINE configs[2] (the headline); network = BASELINE configs[4]: random network, "
                         "advanced one-to-all, sources dealt over the GPUs (a step = one batch of --net-batch sources)")
    ap.add_argument("--net-batch", type=int, default=32, help="--workload network: one-to-all sources per step")
    ap.add_argument("--network-n", type=int, default=5000000, help="nodes of the config4_network leg (BASELINE: 5e6)")
    ap.add_argument("--geometric-n", type=int, default=1000000, help="nodes of the network_geometric leg")
    ap.add_argument("--cpu-full-size", default="auto", choices=["auto", "0", "1"],
                    help="cpu_baseline.value MEASURED at the full size in this run (a second CPU child, in the background "
                         "while the GPU legs run): auto = when the host has the memory (>= 200 GB available) and >= 32 cores")
    ap.add_argument("--cpu-full-only", type=int, default=0, help="internal (CPU child): measure the oracle at --size")
    ap.add_argument("--cpu-full-threads", type=int, default=16)
    args = ap.parse_args()
    if args.cpu_full_only > 0:
        print(json.dumps(cpu_full_size_measure(args.size, args.cpu_full_threads, args.precision == "single")), flush=True)
        return
    if args.cpu_baseline_only > 0:
        # CPU child: the cpu_baseline object (+ the tight resistances of the headline's parity figure) and, when asked for,
        # the CHECKER side of the other legs' parity figures -- everything that touches oracle/ lives in this process
        res = cpu_baseline_entry(cpu_baseline(args.cpu_sample, single=args.precision == "single"),
                                 args.cpu_baseline_only, args.size, args.cpu_sample)
        legs = {}
        for name in [x for x in args.cpu_legs.split(",") if x]:
            try:
                if name == "nodata":
                    legs[name] = oracle_leg_nodata(args.leg_sample)
                elif name == "fp32":
                    legs[name] = oracle_leg_fp32(args.leg_sample)
                elif name == "geometric":
                    legs[name] = oracle_leg_geometric(args.geometric_n)
            except Exception as e:
                legs[name] = {"failed": repr(e)}
        res["_legs"] = legs
        print(json.dumps(res), flush=True)
        return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    import torch
    ndev = max(torch.cuda.device_count(), 1)
    dev_index = local_rank % ndev  # one rank per GPU under the driver; ranks share a GPU only in the gloo self-test
    if world > 1:
        import torch.distributed as dist
        if torch.cuda.is_available():
            torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend="gloo")
    dev = torch.device("cuda", dev_index)

    import circuitscape_jl_amd  # noqa: F401
    from circuitscape_jl_amd import lib
    lib.load(os.environ.get("CSGPU_LIB"))  # default: the in-tree hipcc build; fails loudly if it is missing
    if lib.device_count() < 1:
        raise SystemExit("no HIP device visible")

    dtype = np.float64 if args.precision == "double" else np.float32
    vb = 8 if dtype == np.float64 else 4
    size = args.size
    cells, pairs = focal_pairs(size)
    extra = {}
    for kv in args.opt:
        k, v = kv.split("=")
        extra[k] = float(v) if k in ("theta", "omega_p", "omega_s", "rtol", "atol") else int(v)
    B = args.batch
    K, Wm = args.steps, args.warmup

    def make_opts(precond, **kw):
        o = dict(extra)
        o.update(kw)
        return lib.default_opts(device=dev_index, batch=o.pop("batch", B), criterion=args.criterion,
                                precond_bytes=4 if (precond == "fp32" and vb == 8) else 0, **o)

    has_cuda = torch.cuda.is_available()  # False only in the CPU self-test (emulator library via CSGPU_LIB)

    def sync():
        if has_cuda:
            torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            if has_cuda:
                torch.cuda.synchronize(dev)

    def batch_pairs(step_index):
        # rank r takes batches r, r+world, ... of the (cyclic) lexicographic pair list
        b = step_index * world + rank
        idx = [(b * B + c) % len(pairs) for c in range(B)]
        return [pairs[i][0] for i in idx], [pairs[i][1] for i in idx]

    if args.workload == "network":
        network_workload(args, lib, torch, dist, dev, rank, world, sync)
        return
    if args.scaling == "strong":
        strong_scaling(args, lib, torch, dist, dev, rank, world, make_opts, dtype, vb, pairs, sync)
        return

    g = make_raster(size, dtype=dtype)
    # untimed warm-up of the setup path on a small raster (first launch of every kernel loads its code object; a fresh
    # process pays ~1 s for that once -- the solve path is warmed by the --warmup batches below)
    other = {"same": "fp32", "fp32": "same"}[args.precond] if vb == 8 else None
    for precond in ((args.precond, other) if other else (args.precond,)):
        wn = min(768, size)
        hw = lib.raster_setup(np.ascontiguousarray(g[:wn, :wn]), make_opts(precond))
        hw.solve_pairs([0] * B, [wn * wn - 1] * B)
        hw.close()
    # The first full-size setup of a process also pays the driver for ~100 GB of fresh device memory (page tables: ~1 s,
    # profiles/r2_alloc_probe_*.jsonl); the library keeps released blocks in a pool, so every later setup of the process
    # -- the reference factorises once per component / focal region / source -- runs at the speed reported as `setup_s`.
    # Both are measured: cold = first setup, warm = the same setup again after closing the first handle.
    t0 = time.time()
    h = lib.raster_setup(g, make_opts(args.precond))
    t_setup_cold_wall = time.time() - t0
    cold = h.info
    h.solve_pairs(*batch_pairs(0))          # work vectors of a batch are part of what the pool must hold
    h.close()
    t0 = time.time()
    h = lib.raster_setup(g, make_opts(args.precond))
    t_setup_wall = time.time() - t0
    info = h.info
    one_call = args.calls == "single"
    elapsed, results, agg = run_pairs(h, batch_pairs, K, Wm, sync, one_call=one_call)
    res_local = torch.from_numpy(np.concatenate(results).astype(np.float64))
    res_local = res_local.to(dev) if (has_cuda and (dist is None or args.backend == "nccl")) else res_local
    t1 = time.perf_counter()
    if dist is not None:
        gathered = [torch.empty_like(res_local) for _ in range(world)]
        dist.all_gather(gathered, res_local)  # the path's only collective: final result gather over RCCL/xGMI
    sync()
    t_gather = time.perf_counter() - t1
    elapsed_rank = elapsed + t_gather
    elapsed = elapsed_rank
    rank_reports = [rank_identity(torch, rank, dev_index, has_cuda, elapsed_rank / K * 1e3, K * B, agg)]
    multi = None
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        # what lets a reader check an N-GPU line without trusting it (outside the timed region): the process group's own
        # idea of its size and backend, every rank's device ordinal / PCI bus id / step time / pairs, the gather's size
        rank_reports = [None] * world
        dist.all_gather_object(rank_reports, rank_identity(torch, rank, dev_index, has_cuda, elapsed_rank / K * 1e3,
                                                           K * B, agg))
        allR = torch.cat([t_.cpu() for t_ in gathered]).numpy()
        multi = {"rccl_world_size": int(dist.get_world_size()), "dist_backend": str(dist.get_backend()),
                 "ranks": rank_reports, "distinct_devices": len({(r_["host"], r_["pci_bus_id"] or r_["device_ordinal"])
                                                                 for r_ in rank_reports}),
                 "gather": {"collective": "all_gather", "bytes_per_rank": int(res_local.numel() * res_local.element_size()),
                            "bytes_total": int(res_local.numel() * res_local.element_size() * world),
                            "seconds_incl_barrier": t_gather, "pairs_received": int(allR.size),
                            "all_finite": bool(np.all(np.isfinite(allR)))}}

    def roofline_of(info_p, agg_p):
        """roofline object of the fine-level CG product of one path: algorithmic bytes per launch (reported by the
        library) / mean HIP-event duration of those launches on the library's stream, measured live in this run"""
        avg_ms = agg_p["cg_spmv_ms"] / max(agg_p["cg_spmv_calls"], 1)
        nbytes = agg_p["cg_spmv_bytes"]
        ach = nbytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic, tnote = pmc_traffic(size, B, vb, info_p["lattice_period"] > 0, info_p["precond_bytes"] or vb)
        return {"bound": "hbm", "kernel": cg_product_name(info_p, B, vb), "achieved": ach, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tnote,
                "algorithmic_bytes_per_launch": nbytes, "avg_ms": avg_ms, "launches_timed": agg_p["cg_spmv_calls"]}

    def resid_roofline_of(info_p, agg_p):
        """roofline object of the OTHER big launch of an iteration on the lattice path: the residual update r -= alpha (A p)
        with A p recomputed from the lattice form -- since round 6 fused with the restriction b_c = Q^T r of the V-cycle in
        one marching pass (csrc/lattice.h) when the whole solve runs in double precision. Same events, same iterations as
        the CG product's; algorithmic bytes from the library (csgpu_stats.resid_bytes, formula in include/csgpu.h)."""
        if not agg_p.get("resid_calls") or not agg_p.get("resid_bytes"):
            return None
        tn = {8: "double", 4: "float"}
        avg_ms = agg_p["resid_ms"] / max(agg_p["resid_calls"], 1)
        nbytes = agg_p["resid_bytes"]
        ach = nbytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        xb = info_p["precond_bytes"] or vb
        if agg_p.get("resid_fused"):
            name = ("lattice_rupd_restrict_kernel<%s,%d,512> (residual update with A p recomputed + restriction of the V-cycle, "
                    "one marching pass)" % (tn[vb], B))
            prefix = "void csgpu::lattice_rupd_restrict_kernel<%s, %d, 512, false>" % (tn[vb], B)
        else:
            name = "dia_cg_kernel<%s,%s,%d,RUPD> (residual update with A p recomputed)" % (tn[vb], tn[xb], B)
            prefix = "void csgpu::dia_cg_kernel<%s, %s, %d, 3>" % (tn[vb], tn[xb], B)
        return {"bound": "hbm", "kernel": name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": None, "traffic_source": "not collected", "algorithmic_bytes_per_launch": nbytes, "avg_ms": avg_ms,
                "launches_timed": agg_p["resid_calls"], "_prefix": prefix}

    if rank == 0:
        pairs_done = K * B * world
        setup_s = (info["setup_ms"] + info["upload_ms"]) / 1e3
        # setup is per GPU and amortised over the config's 100 pairs per matrix
        value = pairs_done / (elapsed + setup_s * (K * B) / 100.0)
        spmv1_ms = h.spmv_bench(1, 10)
        mixed = vb == 8 and info["precond_bytes"] == 4
        path_name = "mixed" if mixed else ("fp64" if vb == 8 else "fp32")
        roof = roofline_of(info, agg)
        roof.update({"spmv_k1_avg_ms": spmv1_ms,
                     "spmv_k1_GBs": info["spmv_bytes_fine"] / (spmv1_ms * 1e-3) / 1e9 if spmv1_ms > 0 else 0.0})
        tn_ = {8: "double", 4: "float"}
        roof["_prefix"] = "void csgpu::dia_cg_kernel<%s, %s, %d, 1>" % (tn_[vb], tn_[info["precond_bytes"] or vb], B)
        # the DOMINANT kernel is the one with the longest launch: since the residual update carries the restriction that is
        # the fused pass, not the CG product (which stays beside it as roofline_cg_product)
        roof_resid = resid_roofline_of(info, agg)
        roof_cg = roof
        if roof_resid is not None and roof_resid["avg_ms"] > roof["avg_ms"]:
            roof_resid["dominant_by"] = "%.2f ms per launch against %.2f ms of the CG product, one launch each per iteration" % (
                roof_resid["avg_ms"], roof["avg_ms"])
            roof_resid.update({"spmv_k1_avg_ms": roof["spmv_k1_avg_ms"], "spmv_k1_GBs": roof["spmv_k1_GBs"]})
            roof = roof_resid
        out = {
            "metric": "pair-solves/sec (AMG-PCG, setup amortised over 100 pairs) on %dx%d raster pairwise" % (size, size),
            "value": value,
            "unit": "pair-solves/s",
            "n_gpus": world,
            "steps": K,
            "warmup": Wm,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64/f32-mixed" if mixed else ("f64" if dtype == np.float64 else "f32"),
            "dtype_note": ("`value` is timed on the %s path. fp64 path: every operation in f64 -- the reference's arithmetic "
                           "(T = Float64, src/run.jl:29, src/core.jl:639). mixed path: CG iteration (x, r, A p, every dot "
                           "product, the residual check) in f64, AMG preconditioner (hierarchy + V-cycle) and the stored "
                           "search direction in f32. The other path runs the same workload, warm-up and step count and is "
                           "reported as value_%s with its own roofline" % (path_name, "fp64" if mixed else "mixed")
                           if vb == 8 else "uniform precision"),
            "data": "synthetic",
            "config": {"workload": "%dx%d synthetic raster, 8-neighbour, %d pairs/GPU in batches of %d, %s"
                                   % (size, size, K * B, B, "fp64" if vb == 8 else "fp32"),
                       "n": info["n"], "nnz": info["nnz"], "batch": B, "levels": info["levels"],
                       "operator_complexity": info["operator_complexity"], "criterion": args.criterion,
                       "preconditioner_precision": "fp32" if info["precond_bytes"] == 4 else "fp64",
                       "cg_product": "lattice form, period %d" % info["lattice_period"] if info["lattice_period"] else "CSR"},
            "solve_only_pairs_per_s": pairs_done / elapsed,
            "setup_s": setup_s, "setup_device_s": info["setup_ms"] / 1e3, "setup_wall_s": t_setup_wall,
            "setup_cold_s": (cold["setup_ms"] + cold["upload_ms"]) / 1e3, "setup_cold_wall_s": t_setup_cold_wall,
            "setup_note": "setup_s = warm (second setup of the process, device memory from the library's pool); "
                          "setup_cold_s = first setup of the process (fresh device memory from the driver)",
            "value_cold_setup": pairs_done / (elapsed + (cold["setup_ms"] + cold["upload_ms"]) / 1e3 * (K * B) / 100.0),
            "iters_mean": agg["total_iters"] / float(K * B), "iters_max": agg["max_iters"],
            "max_relres": agg["max_relres"], "not_converged": agg["not_converged"],
            "pcg_device_ms_per_step": agg["device_ms"] / K,   # HIP-event time of the PCG loops (rest of ms_per_step: host side)
            "ms_per_16_pairs": elapsed / K * 1e3 * 16.0 / B,
            "calls": agg["calls"], "stream": stream_block(agg, K * B, B),
            "roofline": roof,
        }
        if roof is not roof_cg:
            out["roofline_cg_product"] = roof_cg
        elif roof_resid is not None:
            out["roofline_residual_update"] = roof_resid
        out["value_" + path_name] = value
        if multi is not None:
            out["multi_gpu"] = multi
        else:
            out["rank0"] = rank_reports[0]
        if world == 1:
            # BASELINE configs[2] AS A JOB: "100 focal pairs" = the first 100 pairs of the list handed
            # over in ONE csgpu_solve_pairs call on the handle just timed -- 32 + 32 + 32 + 4 at batch 32, the last batch at
            # the width its 4 columns ask for -- plus the set-up. `value` above amortises the set-up over 100 pairs at the
            # pace of FULL batches; value_job is what the config's own job gets.
            try:
                npj = min(100, len(pairs))
                t_j = time.perf_counter()
                Rj, _, _, stj = h.solve_pairs([p_[0] for p_ in pairs[:npj]], [p_[1] for p_ in pairs[:npj]])
                t_j = time.perf_counter() - t_j
                cold_s = (cold["setup_ms"] + cold["upload_ms"]) / 1e3
                out["job_100_pairs"] = {
                    "pairs": npj, "batches": "%d x %d + %d" % (npj // B, B, npj % B), "solve_s": t_j,
                    "job_s": setup_s + t_j, "job_cold_setup_s": cold_s + t_j, "value_job": npj / (setup_s + t_j),
                    "value_job_cold_setup": npj / (cold_s + t_j), "iters_mean": stj["total_iters"] / float(npj),
                    "not_converged": stj["not_converged"], "max_relres": stj["max_relres"],
                    "pcg_device_s": stj["device_ms"] / 1e3,
                    "note": "setup_s (warm) + one call with the config's 100 pairs on the timed handle; the ragged last "
                            "batch runs at its own width (K picked per batch)"}
                out["value_job"] = out["job_100_pairs"]["value_job"]
                out["job_100_pairs_s"] = out["job_100_pairs"]["job_s"]
            except Exception as e:
                out["job_100_pairs"] = {"failed": repr(e)}
        if world == 1 and args.extra_legs:
            try:   # shortcut mode on the handle just timed: 14 anchor solves -> all 105 resistances
                non_anchor = [p for p in pairs if p[0] != int(cells[0])][:B]
                out["shortcut"] = shortcut_leg(h, cells, setup_s, non_anchor)
                out["value_shortcut"] = out["shortcut"]["value"]
            except Exception as e:
                out["shortcut"] = {"failed": repr(e)}
        h.close()
        h = None
        csteps = K if args.compare_steps < 0 else args.compare_steps
        if world == 1 and csteps > 0 and other:
            # the same workload, same warm-up and step count on the other preconditioner precision
            oname = "fp64" if other == "same" else "mixed"
            h2 = lib.raster_setup(g, make_opts(other))      # cold for the block sizes of this precision, see above
            cold2 = h2.info
            h2.solve_pairs(*batch_pairs(0))
            h2.close()
            h2 = lib.raster_setup(g, make_opts(other))
            el2, res2, agg2 = run_pairs(h2, batch_pairs, csteps, Wm, sync, one_call=one_call)
            i2 = h2.info
            s2 = (i2["setup_ms"] + i2["upload_ms"]) / 1e3
            out["value_" + oname] = csteps * B / (el2 + s2 * csteps * B / 100.0)
            ncmp = min(csteps, K)
            out[oname + "_path"] = {
                "value": out["value_" + oname], "solve_only_pairs_per_s": csteps * B / el2, "steps": csteps,
                "ms_per_step": el2 / csteps * 1e3, "setup_s": s2, "setup_cold_s": (cold2["setup_ms"] + cold2["upload_ms"]) / 1e3,
                "setup_device_s": i2["setup_ms"] / 1e3,
                "upload_s": i2["upload_ms"] / 1e3, "iters_mean": agg2["total_iters"] / float(csteps * B),
                "iters_max": agg2["max_iters"], "max_relres": agg2["max_relres"], "not_converged": agg2["not_converged"],
                "pcg_device_ms_per_step": agg2["device_ms"] / csteps, "ms_per_16_pairs": el2 / csteps * 1e3 * 16.0 / B,
                "stream": stream_block(agg2, csteps * B, B),
                "roofline": roofline_of(i2, agg2),
                "max_rel_diff_R_vs_%s_path" % path_name: float(max(np.max(np.abs(res2[k] - results[k]) / np.abs(res2[k]))
                                                                   for k in range(ncmp)))}
            h2.close()
        # The CPU children start HERE -- after the two timed paths, so that the headline figures see an idle host -- and work in
        # the background while the GPU legs below run: (1) the bounded-sample child (cpu_baseline of the task's contract + the
        # checker side of every parity figure), (2) the oracle measured AT THE FULL SIZE when the host
        # has the cores and the memory. Both are collected at the end of the run.
        cpu_children = {}
        if args.cpu_sample > 0 and world == 1:
            import subprocess
            want_legs = "nodata,fp32,geometric" if (args.extra_legs and vb == 8) else ""
            t_children = time.perf_counter()
            cpu_children["sample"] = subprocess.Popen(
                [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", str(info["n"]), "--cpu-sample",
                 str(args.cpu_sample), "--size", str(size), "--precision", args.precision, "--cpu-legs", want_legs,
                 "--leg-sample", str(args.leg_sample), "--geometric-n", str(args.geometric_n)],
                stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            if args.cpu_full_size == "1" or (args.cpu_full_size == "auto" and host_can_measure_full_size(size)):
                cpu_children["full"] = subprocess.Popen(
                    [sys.executable, os.path.abspath(__file__), "--cpu-full-only", "1", "--size", str(size), "--precision",
                     args.precision, "--cpu-full-threads", str(args.cpu_full_threads)],
                    stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if world == 1 and args.extra_legs:
            # what a maps-on run pays on the solve side: the whole solution vector carried (x += alpha p fused into the
            # residual update) and the reference's 1e-4 check evaluated as ||A x - b|| / ||b|| with an explicit product
            # (opts.explicit_check = 1; csgpu_solve_pairs without volt_out otherwise carries x at the focal nodes only)
            try:
                # (batches of at most 16 here: x, A p and b carried for all columns are 3 more n x K vectors)
                hv = lib.raster_setup(g, make_opts(args.precond, explicit_check=1, batch=min(B, 16)))
                vsteps = max(1, min(K, 3))
                wv = min(1, Wm)   # (one warm-up step; the timed steps are the first `vsteps` steps of the main run)
                elv, resv, aggv = run_pairs(hv, batch_pairs, vsteps, wv, sync, first_batch=Wm - wv, one_call=one_call)
                iv = hv.info
                sv = (iv["setup_ms"] + iv["upload_ms"]) / 1e3
                hv.close()
                out["value_with_voltages"] = vsteps * B / (elv + sv * vsteps * B / 100.0)
                out["with_voltages"] = {"value": out["value_with_voltages"], "steps": vsteps, "ms_per_step": elv / vsteps * 1e3,
                                        "iters_mean": aggv["total_iters"] / float(vsteps * B), "max_relres_explicit": aggv["max_relres"],
                                        "not_converged": aggv["not_converged"],
                                        "max_rel_diff_R_vs_value_path": float(max(np.max(np.abs(resv[k] - results[k]) / np.abs(results[k]))
                                                                                  for k in range(min(vsteps, K)))),
                                        "note": "x carried over all n rows + explicit ||Ax-b||/||b|| check; no D2H of voltages"}
            except Exception as e:
                out["with_voltages"] = {"failed": repr(e)}
        leg_seconds = {}
        if world == 1 and args.extra_legs:
            t_leg = time.perf_counter()
            try:   # the same raster with 15 % NODATA cells (cell space, refined tiles, 25-point coarse levels)
                out["nodata15"] = nodata_leg(lib, g, B, make_opts, args.precond, sync)
            except Exception as e:
                out["nodata15"] = {"failed": repr(e)}
            leg_seconds["nodata15"] = time.perf_counter() - t_leg
        if world == 1 and args.extra_legs and vb == 8:
            # the two BASELINE configs the headline does not cover, driver-run: configs[3] precision (fp32) on this raster,
            # configs[4] (network, advanced one-to-all) at its stated size
            t_leg = time.perf_counter()
            try:
                out["config3_fp32"] = config3_fp32_leg(lib, size, B, dev_index, batch_pairs, sync, one_call)
            except Exception as e:
                out["config3_fp32"] = {"failed": repr(e)}
            leg_seconds["config3_fp32"] = time.perf_counter() - t_leg
            t_leg = time.perf_counter()
            try:
                out["config4_network"] = config4_network_leg(lib, dev_index, n=args.network_n, torch=torch if has_cuda else None,
                                                             dev=dev if has_cuda else None)
            except Exception as e:
                out["config4_network"] = {"failed": repr(e)}
            leg_seconds["config4_network"] = time.perf_counter() - t_leg
        if world == 1 and args.host_csr:
            try:
                out.update(host_csr_setup(lib, g, make_opts(args.precond)))
            except Exception as e:  # never at the expense of the line
                out["setup_host_csr_s"] = None
                out["setup_host_csr"] = {"failed": repr(e)}
        del g
        if world == 1 and args.pmc_live and has_cuda and info["lattice_period"] > 0:
            # live HBM traffic of the roofline kernel (the handles of this process are closed; its pooled blocks go back to
            # the driver first so that the child finds the device empty)
            t_leg = time.perf_counter()
            try:
                lib.trim_memory()
                objs = [out[k] for k in ("roofline", "roofline_cg_product", "roofline_residual_update") if k in out and out[k].get("_prefix")]
                got = live_pmc_traffic_many(args, args.precond, [o["_prefix"] for o in objs])
                for o in objs:
                    traffic, note = got[o["_prefix"]]
                    if traffic is not None:
                        o["traffic_committed_pass"] = o["traffic"]
                        o["traffic"] = traffic
                        o["traffic_source"] = note
                        o["traffic_over_algorithmic"] = traffic / max(o["algorithmic_bytes_per_launch"], 1)
                    else:
                        o["traffic_live_failed"] = note
            except Exception as e:
                out["roofline"]["traffic_live_failed"] = repr(e)
            leg_seconds["pmc_live"] = time.perf_counter() - t_leg
        if "sample" in cpu_children:
            try:
                t_leg = time.perf_counter()
                child_out, _ = cpu_children["sample"].communicate(timeout=1200)
                leg_seconds["cpu_child_wait"] = time.perf_counter() - t_leg
                leg_seconds["cpu_child"] = time.perf_counter() - t_children
                cb = json.loads(child_out.strip().splitlines()[-1])
                tight = cb.pop("_tight", None)
                legs = cb.pop("_legs", {})
                out["cpu_baseline"] = cb
                if "full" in cpu_children:
                    # the figure measured at the full size leads; the bounded sample's extrapolation stays beside it
                    try:
                        t_leg = time.perf_counter()
                        full_out, _ = cpu_children["full"].communicate(timeout=900)
                        leg_seconds["cpu_full_size_wait"] = time.perf_counter() - t_leg
                        leg_seconds["cpu_full_size"] = time.perf_counter() - t_children
                        m = json.loads(full_out.strip().splitlines()[-1])
                        cb["bounded_sample"] = {"value": cb["value"], "cores": cb["cores"], "sample": cb["sample"],
                                                "kind": "port, extrapolated linearly in n from the bounded sample"}
                        cb.update({"value": m["value_pair_solves_per_s"], "cores": m["threads"],
                                   "kind": "port", "measured": "at the full size, in this run",
                                   "sample": "oracle (C++ restatement of the reference CG+AMG path) AT THE FULL SIZE of the workload "
                                             "(%dx%d, n = %d, nnz = %d) on this box's host cores, in the background of this run: "
                                             "graph %.0f s, set-up %.0f s, then %d pairs at the reference's tolerances on %d "
                                             "threads (one pair per thread, src/core.jl:262-272) in %.0f s, %s iterations; "
                                             "set-up amortised over 100 pairs" % (size, size, m["n"], m["nnz"], m["graph_build_s"],
                                                                                  m["setup_s"], m["pairs"], m["threads"],
                                                                                  m["pairs_wall_s"], m["iters"]),
                                   "host_cores": m["host_cores"], "full_size": m})
                    except Exception as e:
                        try:
                            cpu_children["full"].kill()
                        except Exception:
                            pass
                        cb["kind"] = "port (extrapolated from the bounded sample: the full-size measurement failed: %r)" % (e,)
                else:
                    cb["kind"] = "port (extrapolated linearly in n from the bounded sample; host too small for the full size)" \
                        if size >= 8000 else "port"
                if tight:
                    out["parity"] = gpu_parity(lib, args.cpu_sample, tight, make_opts, dtype, vb == 8)
                # an oracle figure on every leg: the leg's own options on a bounded sample of the leg's
                # own workload against the tight oracle on the HOST-built graph of that sample
                t_leg = time.perf_counter()
                for leg, key, what in (("nodata15", "nodata", "%dx%d raster of the bench generator, 15 %% NODATA (the leg's mask generator)"),
                                       ("config3_fp32", "fp32", "%dx%d raster of the bench generator in fp32, reference regularisation")):
                    tl = legs.get(key)
                    if not tl or leg not in out or "failed" in out[leg]:
                        continue
                    if "failed" in tl:
                        out[leg]["parity"] = {"failed": tl["failed"]}
                        continue
                    try:
                        ns_ = tl["sample_size"]
                        if key == "nodata":
                            gs = nodata_raster(make_raster(ns_, dtype=dtype))
                            out[leg]["parity"] = leg_parity(lib, gs, make_opts(args.precond), tl, 1e-6, what % (ns_, ns_))
                        else:
                            gs = make_raster(ns_, dtype=np.float32)
                            out[leg]["parity"] = leg_parity(lib, gs, lib.default_opts(device=dev_index, batch=B), tl, 1e-4,
                                                            what % (ns_, ns_))
                    except Exception as e:
                        out[leg]["parity"] = {"failed": repr(e)}
                tl = legs.get("geometric")
                if tl and "failed" not in tl:
                    try:
                        out["network_geometric"] = network_geometric_leg(lib, dev_index, tl, n=args.geometric_n,
                                                                         torch=torch if has_cuda else None,
                                                                         dev=dev if has_cuda else None)
                    except Exception as e:
                        out["network_geometric"] = {"failed": repr(e)}
                elif tl:
                    out["network_geometric"] = {"failed": tl["failed"]}
                leg_seconds["leg_parity_gpu"] = time.perf_counter() - t_leg
            except Exception as e:  # the GPU line must be printed whatever happens to the host-side leg
                out.setdefault("cpu_baseline", {"value": None, "unit": "pair-solves/s", "cores": 0, "kind": "port",
                                                "sample": "failed: %r" % (e,)})
        out["leg_seconds"] = leg_seconds
        for k_ in ("roofline", "roofline_cg_product", "roofline_residual_update"):
            if isinstance(out.get(k_), dict):
                out[k_].pop("_prefix", None)
        print(json.dumps(out), flush=True)
    try:
        if h is not None:
            h.close()
    except Exception:
        pass
    if dist is not None:
        dist.destroy_process_group()


def gpu_parity(lib, sample_size, tight, make_opts, dtype, both):
    """The GPU paths, with exactly the options the timed runs used, on the CPU leg's sample raster against the TIGHT
    oracle's resistances of the same pairs (a parity figure accompanies every number)."""
    g = make_raster(sample_size, dtype=dtype)
    cells, pairs = focal_pairs(sample_size)
    nt = tight["pairs"]
    src = [p[0] for p in pairs[:nt]]
    dst = [p[1] for p in pairs[:nt]]
    Ro = np.asarray(tight["R"])
    # north_star tolerance for fp64; fp32 handles (BASELINE configs[3] precision) are held to 1e-4 against the tight oracle
    # on the same fp32-shifted matrix ("the fp32 contract"; measured 3e-7; the reference's own
    # single-precision tolerance is 1e-2 absolute, test/test_utils.jl:72-73)
    out = {"n": sample_size * sample_size, "pairs": nt, "oracle": "tight (true-residual rtol 1e-12)",
           "oracle_max_true_relres": tight["max_true_relres"], "tolerance": 1e-6 if dtype == np.float64 else 1e-4}
    for name, precond in ((("fp64", "same"), ("mixed", "fp32")) if both else (("uniform", "same"),)):
        h = lib.raster_setup(g, make_opts(precond))
        R, _, _, st = h.solve_pairs(src, dst)
        h.close()
        out["max_rel_err_vs_oracle_" + name] = float(np.max(np.abs(R - Ro) / np.abs(Ro)))
        out["iters_mean_" + name] = st["total_iters"] / float(nt)
    out["max_rel_err_vs_oracle"] = max(v for k, v in out.items() if k.startswith("max_rel_err_vs_oracle_"))
    out["ok"] = bool(out["max_rel_err_vs_oracle"] < out["tolerance"])
    return out


def host_csr_setup(lib, g, opts):
    """csgpu_setup from host CSR arrays exactly as a Julia host would hand them (SparseMatrixCSC{Float64,Int64}: Int64,
    1-based; src/run.jl:29-34, src/core.jl:164): upload + index conversion + lattice detection + AMG setup. The matrix is
    obtained by downloading the device-built Laplacian of the same raster (host-side graph construction is the
    reference's job and not timed)."""
    h = lib.raster_setup(g, opts)
    A = h.level_matrix(0, "A")
    h.close()
    rp = np.ascontiguousarray(A.indptr.astype(np.int64) + 1)
    ci = np.ascontiguousarray(A.indices.astype(np.int64) + 1)
    va = np.ascontiguousarray(A.data, dtype=g.dtype)
    n, nnz = A.shape[0], A.nnz
    del A
    t0 = time.perf_counter()
    h2 = lib.setup_arrays(rp, ci, va, n, nnz, opts, index_base=1)
    wall = time.perf_counter() - t0
    i2 = h2.info
    res = {"setup_host_csr_s": wall, "setup_host_csr": {"upload_convert_s": i2["upload_ms"] / 1e3,
                                                        "device_setup_s": i2["setup_ms"] / 1e3,
                                                        "host_bytes": int(rp.nbytes + ci.nbytes + va.nbytes),
                                                        "lattice_period_detected": i2["lattice_period"]}}
    h2.close()
    return res


def strong_scaling(args, lib, torch, dist, dev, rank, world, make_opts, dtype, vb, pairs_all, sync):
    """BASELINE configs[2]/[3] as a fixed-size job: --pairs pairs in total, dealt over the ranks at PAIR granularity
    (rank r takes a contiguous slice of ceil/floor(npairs/world) pairs, so every GPU is busy even when there are
    fewer batches than GPUs) and solved in batches of --batch; every rank builds its own copy of the hierarchy; the
    timed region is the whole job: raster upload + graph build + AMG setup + solves + the result gather."""
    from circuitscape_jl_amd import shard
    size, B = args.size, args.batch
    npairs = args.pairs
    plist = [pairs_all[i % len(pairs_all)] for i in range(npairs)]
    g = make_raster(size, dtype=dtype)
    lo, hi = shard.pair_slice(npairs, rank, world)
    # warm-up: one small problem through the same code path (library load, kernel code objects, allocator)
    wn = min(768, size)
    hw = lib.raster_setup(g[:wn, :wn].copy(), make_opts(args.precond))
    hw.solve_pairs([0] * B, [wn * wn - 1] * B)
    hw.close()
    # The job runs TWICE in this process. The first run pays the driver for ~100 GB of fresh device memory (page tables:
    # seconds, profiles/r2_alloc_probe_*.jsonl -- 7.4 s against 2.4 s for the 100-pair fp64 job on one GPU); the second
    # finds the blocks in the library's pool, which is the state of any process that solves more than one problem. `value`
    # and the speed-up figures are the second run's; the first is reported as job_cold_s.
    job_cold_s = None
    for attempt in range(2):
        sync()
        t0 = time.perf_counter()
        h = lib.raster_setup(g, make_opts(args.precond))
        t_setup = time.perf_counter() - t0
        R = np.zeros(0)
        st = {"total_iters": 0, "max_relres": 0.0, "not_converged": 0}
        if hi > lo:
            R, _, _, st = h.solve_pairs([p[0] for p in plist[lo:hi]], [p[1] for p in plist[lo:hi]])
        t_busy = time.perf_counter() - t0
        full = shard.gather_pairs(np.asarray(R, dtype=np.float64), np.arange(lo, hi), npairs, dist,
                                  dev if (dist is None or args.backend == "nccl") else None)
        sync()
        elapsed = time.perf_counter() - t0
        info = h.info
        h.close()
        if attempt == 0:
            job_cold_s = elapsed
    busy = [t_busy]
    setups = [t_setup]
    has_cuda = torch.cuda.is_available()
    dev_index = dev.index if dev is not None and dev.index is not None else 0
    agg = {"total_iters": st["total_iters"], "max_relres": st["max_relres"], "not_converged": st["not_converged"]}
    nb_mine = max(1, -(-(hi - lo) // B))
    reports = [rank_identity(torch, rank, dev_index, has_cuda, (t_busy - t_setup) / nb_mine * 1e3, hi - lo, agg)]
    multi = None
    if dist is not None:
        t = torch.tensor([elapsed, t_busy, t_setup], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        allt = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        elapsed = max(float(x[0]) for x in allt)
        busy = [float(x[1]) for x in allt]
        setups = [float(x[2]) for x in allt]
        reports = [None] * world
        dist.all_gather_object(reports, rank_identity(torch, rank, dev_index, has_cuda, (t_busy - t_setup) / nb_mine * 1e3,
                                                      hi - lo, agg))
        slot = max(-(-npairs // world), 1)
        multi = {"rccl_world_size": int(dist.get_world_size()), "dist_backend": str(dist.get_backend()), "ranks": reports,
                 "distinct_devices": len({(r_["host"], r_["pci_bus_id"] or r_["device_ordinal"]) for r_ in reports}),
                 "gather": {"collective": "all_gather of (index, value) rows", "bytes_per_rank": slot * 16,
                            "bytes_total": slot * 16 * world, "pairs_received": int(np.sum(~np.isnan(full)))}}
    if rank == 0:
        per_batch = (busy[0] - setups[0]) / max(1, -(-(hi - lo) // B))
        nb1 = -(-npairs // B)
        nbN = -(-(-(-npairs // world)) // B)
        out = {
            "metric": "pair-solves/sec, whole job (%d pairs, setup included) on %dx%d raster pairwise" % (npairs, size, size),
            "value": npairs / elapsed, "unit": "pair-solves/s", "n_gpus": world, "steps": nbN, "warmup": 0,
            "ms_per_step": elapsed / max(nbN, 1) * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64" if vb == 8 else "f32", "data": "synthetic",
            "config": {"workload": "%dx%d synthetic raster, 8-neighbour, %d pairs in total over %d GPU(s), batches of %d"
                                   % (size, size, npairs, world, B), "n": info["n"], "nnz": info["nnz"], "batch": B,
                       "preconditioner_precision": "fp32" if info["precond_bytes"] == 4 else "fp64"},
            "job_s": elapsed, "job_cold_s": job_cold_s, "rank_busy_s": busy, "rank_setup_s": setups, "pairs_per_rank": -(-npairs // world),
            "per_batch_s_rank0": per_batch,
            # predicted: T_1 / T_N = (s + nb_1 b) / (s + nb_N b) from rank 0's measured setup s and per-batch time b;
            # achieved: the one-GPU time RECONSTRUCTED from this run's own pieces -- one setup plus every rank's solve
            # time back to back -- over the measured job time (an N = 1 run of the same command gives the true T_1)
            "predicted_speedup_vs_1gpu": (setups[0] + nb1 * per_batch) / (setups[0] + nbN * per_batch),
            "achieved_speedup_vs_1gpu_reconstructed": (setups[0] + sum(b_ - s_ for b_, s_ in zip(busy, setups))) / elapsed,
            "max_relres": st["max_relres"], "all_pairs_gathered": bool(not np.any(np.isnan(full))),
        }
        if multi is not None:
            out["multi_gpu"] = multi
        else:
            out["rank0"] = reports[0]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def network_workload(args, lib, torch, dist, dev, rank, world, sync):
    """BASELINE configs[4] -- "Network-mode 50M-edge random Laplacian, advanced one-to-all, fp64, 8 x MI355X" -- as the
    driver's command runs it: `bench.py --workload network --gpus N --steps K --warmup W`. Every rank builds the SAME random
    network (5e6 nodes, 5e7 undirected edges; seeded generator) and its own replica of the handle ("replicas only across
    sources"); a step = one batch of --net-batch one-to-all sources (unit current into one focal node, the
    other focal nodes of the batch's focal set tied to ground: src/raster/onetoall.jl:106-117 -> multiple_solver,
    src/raster/advanced.jl:274-312) through csgpu_solve_sources with the sources' voltages and the cumulative node-current
    vector kept; the job's N * K batches are dealt over the ranks as contiguous slices (shard.solve_sources_sharded); the
    timed region ends with the path's two collectives -- ONE all_gather of the sources' voltages and ONE all_reduce(SUM) of
    the cumulative current vector (the serial merge of onetoall.jl:153-158; 40 MB at n = 5e6). Weak scaling: K batches per
    GPU whatever N. Parity: rank 0 solves its first two columns again with voltages back and checks them against scipy's
    Jacobi-CG on the reduced system."""
    from circuitscape_jl_amd import shard
    has_cuda = torch.cuda.is_available()
    dev_index = dev.index if dev is not None and dev.index is not None else 0
    B, K, Wm = args.net_batch, args.steps, args.warmup
    t0 = time.perf_counter()
    G, rng = random_network(args.network_n, torch=torch if has_cuda else None, dev=dev if has_cuda else None)
    t_gen = time.perf_counter() - t0
    n = G.shape[0]
    # the focal set: B nodes per batch, (W + K) * world batches; a batch's columns ground the batch's other focal nodes
    nb_total = (Wm + K) * world
    focal = rng.choice(n, size=nb_total * B, replace=False).reshape(nb_total, B)
    src, grounds, chk = [], [], []
    for b in range(nb_total):
        s_, g_, c_ = one_to_all_columns(focal[b])
        src += s_
        grounds += g_
        chk += c_
    t0 = time.perf_counter()
    h = lib.setup(G, lib.default_opts(device=dev_index, batch=B, precond_bytes=4, itmax=2000), index_dtype=np.int32,
                  index_base=0)
    t_setup = time.perf_counter() - t0
    info = h.info
    gdev = dev if (has_cuda and (dist is None or args.backend == "nccl")) else None
    nw = Wm * world * B
    if nw > 0:   # warm-up batches: same code path incl. the collectives
        shard.solve_sources_sharded(h, src[:nw], grounds[:nw], check=chk[:nw], dist=dist, device=gdev, want_cum=True)
    sync()
    t0 = time.perf_counter()
    v, cum, _, stats = shard.solve_sources_sharded(h, src[nw:], grounds[nw:], check=chk[nw:], dist=dist, device=gdev,
                                                   want_cum=True)
    t_rank = time.perf_counter() - t0
    sync()
    elapsed = time.perf_counter() - t0
    st = stats[0] if stats else {"total_iters": 0, "max_iters": 0, "max_relres": 0.0, "not_converged": 0, "device_ms": 0.0,
                                 "cg_spmv_ms": 0.0, "cg_spmv_calls": 0, "cg_spmv_bytes": 0, "batch": B}
    agg = {"total_iters": st["total_iters"], "max_relres": st["max_relres"], "not_converged": st["not_converged"]}
    reports = [rank_identity(torch, rank, dev_index, has_cuda, t_rank / max(K, 1) * 1e3, K * B, agg)]
    multi = None
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        reports = [None] * world
        dist.all_gather_object(reports, rank_identity(torch, rank, dev_index, has_cuda, t_rank / max(K, 1) * 1e3, K * B, agg))
        multi = {"rccl_world_size": int(dist.get_world_size()), "dist_backend": str(dist.get_backend()), "ranks": reports,
                 "distinct_devices": len({(r_["host"], r_["pci_bus_id"] or r_["device_ordinal"]) for r_ in reports}),
                 "collectives": {"all_gather": "(index, voltage) rows, %d B per rank" % (K * B * 16),
                                 "all_reduce_sum": "cumulative node-current vector, %d B" % (n * 8)},
                 "gather": {"pairs_received": int(np.sum(~np.isnan(v)))}}
    if rank == 0:
        nsrc = K * B * world
        out = {
            "metric": "one-to-all sources/sec (PCG on one shared hierarchy, setup amortised over the job) on a %d-node network" % n,
            "value": nsrc / (elapsed + t_setup), "unit": "one-to-all sources/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": elapsed / max(K, 1) * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "random network n = %d, %d undirected edges (BASELINE configs[4]), advanced one-to-all, "
                                   "%d sources/GPU in batches of %d, fp64 (fp32 preconditioner)" % (n, (G.nnz - n) // 2, K * B, B),
                       "n": int(n), "nnz": int(G.nnz), "batch": B, "levels": info["levels"],
                       "preconditioner": "AMG" if info["levels"] > 1 else "Jacobi (expander: the setup declines to coarsen)"},
            "solve_only_sources_per_s": nsrc / elapsed, "setup_s": t_setup, "setup_device_s": info["setup_ms"] / 1e3,
            "generate_s": t_gen, "iters_mean": st["total_iters"] / float(max(K * B, 1)), "iters_max": st["max_iters"],
            "max_relres": st["max_relres"], "not_converged": st["not_converged"],
            "pcg_device_ms_per_step": st["device_ms"] / max(K, 1), "rank0_wall_ms_per_step": t_rank / max(K, 1) * 1e3,
            "value_device_rank0": K * B / max(st["device_ms"] / 1e3, 1e-12),
            "roofline": csr_spmm_roofline(st, info, st["batch"]),
            "all_sources_gathered": bool(not np.any(np.isnan(v))), "cum_current_sum": float(cum.sum()),
        }
        if multi is not None:
            out["multi_gpu"] = multi
        else:
            out["rank0"] = reports[0]
        # parity of this rank's first two columns (untimed): voltages back, scipy Jacobi-CG on the reduced system
        try:
            lo, _ = shard.pair_slice(K * B * world, rank, world)
            cols = [nw + lo, nw + lo + 1][: min(2, K * B)]
            vck, X, _, _ = h.solve_sources([src[c] for c in cols], [grounds[c] for c in cols], check=[chk[c] for c in cols],
                                           want_voltages=True)
            rerr = verr = 0.0
            for k_, c in enumerate(cols):
                b = np.zeros(n)
                b[chk[c]] = 1.0
                xs, flag, res = _masked_jacobi_cg(G, b, grounds[c])
                rerr = max(rerr, abs(v[c - nw] - xs[chk[c]]) / abs(xs[chk[c]]), abs(vck[k_] - xs[chk[c]]) / abs(xs[chk[c]]))
                verr = max(verr, float(np.max(np.abs(X[:, k_] - xs)) / np.max(np.abs(xs))))
            out["parity"] = {"max_rel_err": float(max(rerr, verr)), "tolerance": 1e-6, "ok": bool(max(rerr, verr) < 1e-6),
                             "columns_checked": len(cols), "checker": "scipy CG + Jacobi on the reduced system, true residual 1e-12"}
        except Exception as e:
            out["parity"] = {"failed": repr(e)}
        print(json.dumps(out), flush=True)
    h.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
