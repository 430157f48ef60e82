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
