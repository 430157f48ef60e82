"""Worker for test_dist_gloo.py: one process per rank, gloo backend, kernels on the CPU emulator build."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist
    import circuitscape_jl_amd  # noqa: F401
    from circuitscape_jl_amd import lib, shard
    out_path = sys.argv[1]
    os.environ["HIPEMU_THREADS"] = "2"
    lib.load(os.path.join(ROOT, "tests", "emu", "libcsgpu_emu.so"))
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    N = 36
    g = 1.0 / np.exp(np.random.default_rng(12345).standard_normal((N, N)))
    cells = np.random.default_rng(67890).choice(N * N, size=6, replace=False)
    src = [int(cells[i]) for i in range(6) for j in range(i + 1, 6)]
    dst = [int(cells[j]) for i in range(6) for j in range(i + 1, 6)]
    h = lib.raster_setup(g, lib.default_opts(batch=4))
    full, stats = shard.solve_pairs_sharded(h, src, dst, batch=4, dist=dist)
    mine = shard.shard_batches(len(src), 4, rank, world)
    # scope row N1 across ranks: cumulative / maximum node-current vectors, one all_reduce(SUM) + one all_reduce(MAX)
    R2, cum, mx, _ = shard.solve_pairs_currents_sharded(h, src, dst, batch=4, dist=dist, want_max=True)
    # BASELINE configs[4] across ranks: one-to-all columns dealt as contiguous slices, check voltages gathered, the
    # cumulative / maximum node-current vectors reduced (SUM / MAX)
    pts = [int(c) for c in cells]
    osrc = [[p] for p in pts]
    ognd = [[q for q in pts if q != p] for p in pts]
    v, ocum, omx, ost = shard.solve_sources_sharded(h, osrc, ognd, check=pts, dist=dist, want_cum=True, want_max=True)
    lo, hi = shard.pair_slice(len(pts), rank, world)
    h.close()
    if rank == 0:
        json.dump({"R": full.tolist(), "src": src, "dst": dst, "n_mine_rank0": int(len(mine)), "world": world,
                   "R2": R2.tolist(), "cum": cum.tolist(), "max": mx.tolist(), "pts": pts, "v": v.tolist(),
                   "ocum": ocum.tolist(), "omax": omx.tolist(), "cols_rank0": [lo, hi], "ost_nrhs": ost[0]["nrhs"]},
                  open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
