"""N > 1 path on CPU: world_size 2, gloo backend (the GPU run uses the same code with backend nccl = RCCL)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_batches_partition():
    import circuitscape_jl_amd  # noqa: F401
    from circuitscape_jl_amd import shard
    for npairs, batch, world in [(15, 4, 2), (100, 8, 8), (3, 8, 4), (0, 8, 2), (105, 16, 3)]:
        got = np.concatenate([shard.shard_batches(npairs, batch, r, world) for r in range(world)])
        assert sorted(got.tolist()) == list(range(npairs))
        sizes = [len(shard.shard_batches(npairs, batch, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= batch  # balanced to within one batch (no triangular load)


def test_two_ranks_gloo_match_oracle(emu_lib, oracle, tmp_path):
    out = tmp_path / "dist.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "tests", "_dist_worker.py"), str(out)]
    subprocess.check_call(cmd, env=env, cwd=ROOT, timeout=600)
    d = json.load(open(out))
    assert d["world"] == 2 and 0 < d["n_mine_rank0"] < len(d["src"])
    from oracle import refgraph as rg
    N = 36
    g = 1.0 / np.exp(np.random.default_rng(12345).standard_normal((N, N)))
    A = oracle.regularize(rg.raster_laplacian_from_conductance(g))
    Ro, _, _ = oracle.OracleAMG(A).solve_pairs(d["src"], d["dst"], rtol=1e-12, atol=0.0, criterion=1)
    assert np.max(np.abs(np.array(d["R"]) - Ro) / Ro) < 1e-6
    # current maps reduced across the two ranks == the single-process accumulation over all pairs
    assert np.array_equal(np.array(d["R2"]), np.array(d["R"]))
    h = emu_lib.raster_setup(g, emu_lib.default_opts(batch=4))
    n = N * N
    cum = np.zeros(n)
    mx = np.full(n, -9999.0)
    h.solve_pairs_currents(d["src"], d["dst"], want_currents=False, cum=cum, mx=mx)
    h.close()
    assert np.max(np.abs(np.array(d["cum"]) - cum)) < 1e-12 * cum.max()
    assert np.array_equal(np.array(d["max"]), mx)
    # one-to-all sources across the two ranks (BASELINE configs[4]'s sharding: contiguous slices of the columns, one gather,
    # one reduction per map) == the single-process call on all columns
    pts = d["pts"]
    h = emu_lib.raster_setup(g, emu_lib.default_opts(batch=4))
    ocum = np.zeros(n)
    omx = np.zeros(n)
    v, _, _, st = h.solve_sources([[p] for p in pts], [[q for q in pts if q != p] for p in pts], check=pts, cum=ocum, mx=omx)
    h.close()
    assert d["cols_rank0"] == [0, 3] and d["ost_nrhs"] == 3 and st["nrhs"] == 6
    assert np.max(np.abs(np.array(d["v"]) - v) / v) < 1e-9
    assert np.max(np.abs(np.array(d["ocum"]) - ocum)) < 1e-9 * ocum.max()
    assert np.max(np.abs(np.array(d["omax"]) - omx)) < 1e-9 * omx.max()


def test_pair_slice_and_gather_single_process():
    import circuitscape_jl_amd  # noqa: F401
    from circuitscape_jl_amd import shard
    for npairs, world in [(100, 8), (5, 8), (0, 3), (1000, 8), (7, 7)]:
        sl = [shard.pair_slice(npairs, r, world) for r in range(world)]
        assert sl[0][0] == 0 and sl[-1][1] == npairs and all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
        sizes = [b - a for a, b in sl]
        assert max(sizes) - min(sizes) <= 1          # 100 pairs on 8 GPUs: 13,13,13,13,12,12,12,12 -- every GPU busy
    full = shard.gather_pairs(np.arange(5) * 2.0, np.arange(5), 5)
    assert np.array_equal(full, np.arange(5) * 2.0)


def _run_bench(tmp_path, extra, port):
    """bench.py under torch.distributed.run with two ranks on the gloo backend, kernels on the CPU emulator build: the
    N > 1 code path of the bench exactly as the driver launches it (minus RCCL)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HIPEMU_THREADS="2",
               CSGPU_LIB=os.path.join(ROOT, "tests", "emu", "libcsgpu_emu.so"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--size", "60",
           "--batch", "4", "--cpu-sample", "0"] + extra
    res = subprocess.run(cmd, env=env, cwd=ROOT, timeout=900, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]          # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_two_ranks_weak_scaling_line(emu_lib, tmp_path):
    d = _run_bench(tmp_path, ["--steps", "2", "--warmup", "1"], 29541)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["unit"] == "pair-solves/s" and d["value"] > 0 and d["higher_is_better"] is True
    assert d["config"]["batch"] == 4 and d["not_converged"] == 0 and d["max_relres"] < 1e-4
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["algorithmic_bytes_per_launch"] > 0
    # whole-job aggregate: 2 ranks x 2 steps x 4 pairs
    assert abs(d["solve_only_pairs_per_s"] * d["ms_per_step"] * 1e-3 * 2 - 16) < 1e-6
    _check_multi_gpu_block(d, pairs_per_rank=[8, 8])
    assert d["multi_gpu"]["gather"]["bytes_per_rank"] == 8 * 8 and d["multi_gpu"]["gather"]["all_finite"] is True
    assert d["multi_gpu"]["gather"]["seconds_incl_barrier"] >= 0


def _check_multi_gpu_block(d, pairs_per_rank):
    """what lets a reader verify an N-GPU line without trust -- the process group's own world size and
    backend, and per rank the device ordinal, PCI bus id (None on the CPU emulator), step time and pairs done."""
    m = d["multi_gpu"]
    assert m["rccl_world_size"] == 2 and m["dist_backend"] == "gloo"
    assert [r["rank"] for r in m["ranks"]] == [0, 1]
    for r, want in zip(m["ranks"], pairs_per_rank):
        assert set(r) >= {"host", "device_ordinal", "pci_bus_id", "ms_per_step", "pairs_done", "iters_mean", "max_relres"}
        assert r["pairs_done"] == want and r["ms_per_step"] > 0 and r["not_converged"] == 0
    assert m["gather"]["pairs_received"] == sum(pairs_per_rank)


def test_bench_two_ranks_strong_scaling_line(emu_lib, tmp_path):
    d = _run_bench(tmp_path, ["--scaling", "strong", "--pairs", "11"], 29543)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["all_pairs_gathered"] is True
    assert len(d["rank_busy_s"]) == 2 and d["pairs_per_rank"] == 6 and d["value"] > 0
    _check_multi_gpu_block(d, pairs_per_rank=[6, 5])
    assert d["predicted_speedup_vs_1gpu"] > 1.0 and d["achieved_speedup_vs_1gpu_reconstructed"] > 0.5


def test_bench_two_ranks_network_workload_line(emu_lib, tmp_path):
    """`bench.py --workload network --gpus 2` (BASELINE configs[4] as the driver would launch it, gloo instead of RCCL): the
    sources of the job dealt over the two ranks, the two collectives of the path, the N-GPU verification block, a roofline
    of the CSR SpMM and the scipy parity figure in ONE line from rank 0."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", HIPEMU_THREADS="2",
               CSGPU_LIB=os.path.join(ROOT, "tests", "emu", "libcsgpu_emu.so"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload", "network",
           "--network-n", "4000", "--net-batch", "4", "--steps", "2", "--warmup", "1"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, timeout=900, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["unit"] == "one-to-all sources/s" and d["value"] > 0 and d["not_converged"] == 0 and d["max_relres"] < 1e-4
    assert d["all_sources_gathered"] is True and d["cum_current_sum"] > 0
    assert abs(d["solve_only_sources_per_s"] * d["ms_per_step"] * 1e-3 * 2 - 16) < 1e-6    # 2 ranks x 2 steps x 4 sources
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["algorithmic_bytes_per_launch"] > 0
    assert d["pcg_device_ms_per_step"] > 0 and d["parity"]["ok"] is True and d["parity"]["columns_checked"] == 2
    _check_multi_gpu_block(d, pairs_per_rank=[8, 8])
