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


def pmc_traffic(size, batch, vb):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (tools/gpu_pmc.sh -> profiles/),
    collected with rocprofv3 --pmc in separate passes and corrected as MI355X_MICROARCH.md prescribes. None when no
    profile matches the current workload."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        key = "%d_k%d_f%d" % (size, batch, vb * 8)
        return d.get(key, {}).get("traffic_bytes_per_launch")
    except Exception:
        return None


def make_raster(size, seed=12345, sigma=1.0, dtype=np.float64):
    rng = np.random.default_rng(seed)
    r = np.exp(sigma * rng.standard_normal((size, size)))
    return (1.0 / r).astype(dtype)


def focal_pairs(size, npts=15, seed=67890):
    rng = np.random.default_rng(seed)
    cells = rng.choice(size * size, size=npts, replace=False)
    pairs = [(int(cells[i]), int(cells[j])) for i in range(npts) for j in range(i + 1, npts)]
    return cells, pairs


def cpu_baseline(sample_size, nsolve=2, max_threads=32, ntight=16):
    """Oracle (CPU restatement of the reference CG+AMG path) on a bounded sample of the same workload: one thread, and
    all host cores the way the reference parallelises (one pair per task, src/core.jl:262-272). Also returns the
    TIGHT oracle's resistances (true-residual rtol 1e-12) of the first `ntight` pairs: the parity reference the GPU
    path is compared with on the same sample raster (`parity` in the bench line)."""
    from oracle import refgraph as rg, refsolve as rs
    g = make_raster(sample_size)
    G = rg.raster_laplacian_from_conductance(g)
    A = rs.regularize(G)
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
    if "tight_R" in cb:  # consumed by the parent (parity), not part of the published object
        entry["_tight"] = {"R": cb["tight_R"], "pairs": cb["tight_pairs"], "max_true_relres": cb["tight_max_true_relres"],
                           "wall_s": cb["tight_wall_s"]}
    return entry


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=7)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=10000)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--precision", default="double", choices=["double", "single"])
    ap.add_argument("--cpu-sample", type=int, default=3000, help="raster edge of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--criterion", type=int, default=0)
    ap.add_argument("--precond", default="fp32", choices=["same", "fp32"],
                    help="precision of the AMG preconditioner: fp32 under the fp64 CG iteration (default), or the same "
                         "precision as the CG iteration")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo only to exercise the "
                         "multi-rank code path on a box with fewer GPUs than ranks)")
    ap.add_argument("--opt", action="append", default=[], help="extra csgpu_opts override key=value (tuning)")
    ap.add_argument("--compare-steps", type=int, default=3,
                    help="N=1 only: also time this many steps with an fp64 preconditioner and report them (0 = skip)")
    ap.add_argument("--cpu-baseline-only", type=int, default=0, metavar="N_FULL",
                    help="internal: run only the CPU-baseline leg on a --cpu-sample raster, scale to N_FULL nodes, print "
                         "its JSON object and exit (the bench runs this in a child process so that nothing on the host "
                         "side can take the GPU line down)")
    args = ap.parse_args()
    if args.cpu_baseline_only > 0:
        print(json.dumps(cpu_baseline_entry(cpu_baseline(args.cpu_sample), args.cpu_baseline_only, args.size,
                                            args.cpu_sample)), flush=True)
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
    size = args.size
    g = make_raster(size, dtype=dtype)
    cells, pairs = focal_pairs(size)
    extra = {}
    for kv in args.opt:
        k, v = kv.split("=")
        extra[k] = float(v) if k in ("theta", "omega_p", "omega_s", "rtol", "atol") else int(v)
    opts = lib.default_opts(device=dev_index, batch=args.batch, criterion=args.criterion,
                            precond_bytes=4 if args.precond == "fp32" else 0, **extra)
    t0 = time.time()
    h = lib.raster_setup(g, opts)
    t_setup_wall = time.time() - t0
    del g
    info = h.info
    B = args.batch
    K, Wm = args.steps, args.warmup

    def batch_pairs(step_index):
        # rank r takes batches r, r+world, ... of the (cyclic) lexicographic pair list
        b = step_index * world + rank
        idx = [(b * B + c) % len(pairs) for c in range(B)]
        return [pairs[i][0] for i in idx], [pairs[i][1] for i in idx]

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for w in range(Wm):
        s, d = batch_pairs(w)
        h.solve_pairs(s, d)
    sync()
    t0 = time.perf_counter()
    results = []
    agg = dict(total_iters=0, max_iters=0, cg_spmv_ms=0.0, cg_spmv_calls=0, device_ms=0.0, max_relres=0.0)
    for k in range(K):
        s, d = batch_pairs(Wm + k)
        R, _, _, st = h.solve_pairs(s, d)
        results.append(R)
        agg["total_iters"] += st["total_iters"]
        agg["max_iters"] = max(agg["max_iters"], st["max_iters"])
        agg["cg_spmv_ms"] += st["cg_spmv_ms"]
        agg["cg_spmv_calls"] += st["cg_spmv_calls"]
        agg["device_ms"] += st["device_ms"]
        agg["max_relres"] = max(agg["max_relres"], st["max_relres"])
    res_local = torch.from_numpy(np.concatenate(results).astype(np.float64))
    res_local = res_local.to(dev) if (dist is None or args.backend == "nccl") else res_local
    if dist is not None:
        gathered = [torch.empty_like(res_local) for _ in range(world)]
        dist.all_gather(gathered, res_local)  # the path's only collective: final result gather over RCCL/xGMI
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        pairs_done = K * B * world
        setup_s = (info["setup_ms"] + info["upload_ms"]) / 1e3
        # setup is per GPU and amortised over the config's 100 pairs per matrix
        value = pairs_done / (elapsed + setup_s * (K * B) / 100.0)
        spmv_avg_ms = agg["cg_spmv_ms"] / max(agg["cg_spmv_calls"], 1)
        vb = 8 if dtype == np.float64 else 4
        # CG product y = A p: matrix (values + int32 columns + row pointers) + read p once + write y once; p is stored in
        # the preconditioner's precision (fp32 under the default mixed path), y in the CG precision
        xb = info["precond_bytes"]
        spmm_bytes = info["nnz"] * (vb + 4) + (info["n"] + 1) * 4 + info["n"] * B * (xb + vb)
        achieved = spmm_bytes / (spmv_avg_ms * 1e-3) / 1e9 if spmv_avg_ms > 0 else 0.0
        spmv1_ms = h.spmv_bench(1, 10)
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
            "dtype": "f64" if dtype == np.float64 else "f32",
            "dtype_note": ("CG iteration, residuals, dot products and the residual check in f64; AMG preconditioner "
                           "(hierarchy + V-cycle) in f32" if (vb == 8 and info["precond_bytes"] == 4) else "uniform precision"),
            "data": "synthetic",
            "config": {"workload": "%dx%d synthetic raster, 8-neighbour, %d pairs/GPU in batches of %d, %s"
                                   % (size, size, K * B, B, "fp64" if vb == 8 else "fp32"),
                       "n": info["n"], "nnz": info["nnz"], "batch": B, "levels": info["levels"],
                       "operator_complexity": info["operator_complexity"], "criterion": args.criterion,
                       "preconditioner_precision": "fp32" if info["precond_bytes"] == 4 else "fp64"},
            "solve_only_pairs_per_s": pairs_done / elapsed,
            "setup_s": setup_s, "setup_device_s": info["setup_ms"] / 1e3, "setup_wall_s": t_setup_wall,
            "iters_mean": agg["total_iters"] / float(K * B), "iters_max": agg["max_iters"],
            "max_relres": agg["max_relres"],
            "roofline": {"bound": "hbm", "kernel": "spmv_kernel<%s,%d,PLAIN,DOT,x=%s> (fine-level CG SpMM)" % ("double" if vb == 8 else "float", B, "float" if xb == 4 else "double"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(size, B, vb), "algorithmic_bytes_per_launch": spmm_bytes, "avg_ms": spmv_avg_ms,
                         "launches_timed": agg["cg_spmv_calls"],
                         "spmv_k1_avg_ms": spmv1_ms,
                         "spmv_k1_GBs": info["spmv_bytes_fine"] / (spmv1_ms * 1e-3) / 1e9 if spmv1_ms > 0 else 0.0},
        }
        if world == 1 and args.compare_steps > 0 and args.precond == "fp32" and dtype == np.float64:
            # same workload with the preconditioner in fp64 as well (pure-fp64 path), for comparison
            h.close()
            h2 = lib.raster_setup(make_raster(size, dtype=dtype), lib.default_opts(device=dev_index, batch=B,
                                                                                  criterion=args.criterion))
            s, d = batch_pairs(0)
            h2.solve_pairs(s, d)
            t1 = time.perf_counter()
            its = 0
            for k in range(args.compare_steps):
                s, d = batch_pairs(Wm + k)
                R2, _, _, st2 = h2.solve_pairs(s, d)
                its += st2["total_iters"]
            el2 = time.perf_counter() - t1
            i2 = h2.info
            out["fp64_preconditioner"] = {
                "solve_only_pairs_per_s": args.compare_steps * B / el2,
                "value": args.compare_steps * B / (el2 + (i2["setup_ms"] + i2["upload_ms"]) / 1e3 * args.compare_steps * B / 100.0),
                "steps": args.compare_steps, "iters_mean": its / float(args.compare_steps * B),
                "max_abs_diff_R_vs_fp32_preconditioner": float(np.max(np.abs(R2 - results[args.compare_steps - 1])))
                if args.compare_steps <= K else None}
            h2.close()
        if args.cpu_sample > 0 and world == 1:
            try:
                import subprocess
                child = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", str(info["n"]),
                                        "--cpu-sample", str(args.cpu_sample), "--size", str(size)],
                                       capture_output=True, text=True, timeout=600)
                out["cpu_baseline"] = json.loads(child.stdout.strip().splitlines()[-1])
            except Exception as e:  # the GPU line must be printed whatever happens to the host-side leg
                out["cpu_baseline"] = {"value": None, "unit": "pair-solves/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    try:
        h.close()
    except Exception:
        pass
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
