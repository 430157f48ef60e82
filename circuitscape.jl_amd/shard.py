"""Multi-GPU sharding of the pairwise path (independent pair solves) and of the one-to-all / all-to-one path (independent
sources): replicated matrix + hierarchy, one gather (+ one reduction of the cumulative current map when it is asked for).

The reference has no distributed layer; its only parallelism is one task per source point
(Threads.@spawn, src/core.jl:262-272), which gives a triangular load (core.jl:265-267). Here the unit of work is a
BATCH of `batch` pairs (one multi-RHS solve); batches are dealt round-robin to the ranks (one process per GPU),
every rank holds the whole matrix and AMG hierarchy, and the only collective is the final all_gather of the
per-pair results over RCCL/xGMI (`backend="nccl"` on ROCm) -- or gloo in the CPU tests.
"""
import numpy as np


def shard_batches(npairs, batch, rank, world):
    """Indices (into the global pair list) this rank solves: batches rank, rank+world, ... of `batch` pairs."""
    nb = (npairs + batch - 1) // batch
    mine = []
    for b in range(rank, nb, world):
        mine.extend(range(b * batch, min(npairs, (b + 1) * batch)))
    return np.asarray(mine, dtype=np.int64)


def pair_slice(npairs, rank, world):
    """Contiguous slice [lo, hi) of the global pair list for `rank`: sizes differ by at most one pair, so a job with
    fewer batches than GPUs (100 pairs in batches of 16 on 8 GPUs) still occupies every GPU."""
    base, rem = divmod(npairs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def batch_slot(npairs, batch, world):
    """largest number of pairs shard_batches gives any rank"""
    nb = (npairs + batch - 1) // batch
    return (nb + world - 1) // world * batch


def gather_pairs(values, index, npairs, dist=None, device=None, slot=None):
    """Full per-pair vector on every rank from each rank's (index, value) list: ONE all_gather of fixed-size slots of
    (index, value) rows padded with index -1 (the path's only collective; RCCL over xGMI with backend nccl).
    dist=None: single process. Every pair must have been solved by exactly one rank. `slot` = the largest share a rank
    can hold, known to every rank without communication: ceil(npairs / world) for the contiguous split (pair_slice, the
    default), ceil(nbatches / world) * batch when whole batches are dealt (batch_slot)."""
    full = np.full(npairs, np.nan)
    index = np.asarray(index, dtype=np.int64)
    if dist is None or dist.get_world_size() == 1:
        full[index] = values
    else:
        import torch
        world = dist.get_world_size()
        slot = max(slot if slot is not None else (npairs + world - 1) // world, 1)
        assert len(index) <= slot, (len(index), slot)
        buf = torch.full((slot, 2), -1.0, dtype=torch.float64)
        if len(index):
            buf[: len(index), 0] = torch.from_numpy(index.astype(np.float64))
            buf[: len(index), 1] = torch.from_numpy(np.asarray(values, dtype=np.float64))
        if device is not None:
            buf = buf.to(device)
        out = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(out, buf)
        for t in out:
            rows = t.cpu().numpy()
            ok = rows[:, 0] >= 0
            full[rows[ok, 0].astype(np.int64)] = rows[ok, 1]
    assert not np.any(np.isnan(full)), "some pair was not solved by any rank"
    return full


def solve_pairs_sharded(handle, src, dst, batch, dist=None, device=None):
    """Solve all (src, dst) pairs across the ranks of `dist` (torch.distributed, already initialised) and return the
    full resistance vector on every rank. With dist=None this is a plain single-process solve."""
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    npairs = len(src)
    if dist is None or dist.get_world_size() == 1:
        R, _, _, st = handle.solve_pairs(src, dst)
        return np.asarray(R, dtype=np.float64), [st]
    mine = shard_batches(npairs, batch, dist.get_rank(), dist.get_world_size())
    stats = []
    R = np.zeros(0)
    if len(mine):
        R, _, _, st = handle.solve_pairs(src[mine], dst[mine])
        stats.append(st)
    return gather_pairs(R, mine, npairs, dist, device, slot=batch_slot(npairs, batch, dist.get_world_size())), stats


def solve_pairs_currents_sharded(handle, src, dst, batch, dist=None, device=None, weights=None, want_max=False):
    """Pairwise mode with current maps on (scope row N1) across ranks: every rank accumulates the cumulative
    (and maximum) node-current vector of ITS pairs on its GPU (csgpu_solve_pairs_currents), then the n-vectors are
    combined with ONE all_reduce each -- SUM for the cumulative map, MAX for the maximum map -- which is exactly what
    the reference's serial merge does (`cum.cum_curr[mycsid()] .+= …`, `max.(…)`, src/out.jl:96-107). Resistances are
    gathered as in solve_pairs_sharded. Returns (R, cum, max or None, stats) identical on every rank.
    At n = 1e8 the cumulative vector is 0.8 GB in fp64: one reduction per job, not per pair."""
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    npairs = len(src)
    n = handle.info["n"]
    dt = handle.dtype
    cum = np.zeros(n, dtype=dt)
    mx = np.full(n, -9999.0, dtype=dt) if want_max else None
    w = None if weights is None else np.asarray(weights, dtype=np.int32)
    if dist is None or dist.get_world_size() == 1:
        R, _, _, st = handle.solve_pairs_currents(src, dst, weights=w, want_currents=False, cum=cum, mx=mx)
        return np.asarray(R, dtype=np.float64), cum, mx, [st]
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = shard_batches(npairs, batch, rank, world)
    stats = []
    R = np.zeros(0)
    if len(mine):
        R, _, _, st = handle.solve_pairs_currents(src[mine], dst[mine], weights=None if w is None else w[mine],
                                                  want_currents=False, cum=cum, mx=mx)
        stats.append(st)
    full = gather_pairs(R, mine, npairs, dist, device, slot=batch_slot(npairs, batch, world))
    tc = torch.from_numpy(cum)
    tc = tc.to(device) if device is not None else tc
    dist.all_reduce(tc, op=dist.ReduceOp.SUM)
    cum = tc.cpu().numpy()
    if want_max:
        tm = torch.from_numpy(mx)
        tm = tm.to(device) if device is not None else tm
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        mx = tm.cpu().numpy()
    return full, cum, mx, stats


def solve_sources_sharded(handle, sources, grounds, check=None, values=None, dist=None, device=None, want_cum=False,
                          want_max=False):
    """BASELINE configs[4] across ranks -- advanced one-to-all / all-to-one on one graph: the fan-out over the focal points
    (`Threads.@spawn(f(x))` per point, src/raster/onetoall.jl:146-151, each a multiple_solver call,
    src/raster/advanced.jl:274-312) with one process per GPU. "Replicas only across sources" (SURVEY.md section 8e): every
    rank holds the whole matrix + hierarchy and solves a CONTIGUOUS slice of the columns (pair_slice: sizes differ by at
    most one column) as one csgpu_solve_sources call; the columns' check voltages (`res[i] = v[1]`, onetoall.jl:141) are
    gathered with ONE all_gather, and -- when the cumulative / maximum current map is asked for -- the per-rank n-vectors
    are combined with ONE all_reduce each (SUM / MAX: the serial merge of onetoall.jl:153-158). Returns (check voltages of
    all columns, cum or None, max or None, [stats of this rank]) identical on every rank."""
    nrhs = len(sources)
    n = handle.info["n"]
    dt = handle.dtype
    cum = np.zeros(n, dtype=dt) if want_cum else None
    mx = np.zeros(n, dtype=dt) if want_max else None
    chk = None if check is None else np.asarray(check, dtype=np.int64)
    if dist is None or dist.get_world_size() == 1:
        v, _, _, st = handle.solve_sources(sources, grounds, values=values, check=chk, cum=cum, mx=mx)
        return (None if v is None else np.asarray(v, dtype=np.float64)), cum, mx, [st]
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = pair_slice(nrhs, rank, world)
    stats = []
    v = np.zeros(0)
    if hi > lo:
        v, _, _, st = handle.solve_sources(sources[lo:hi], grounds[lo:hi], values=None if values is None else values[lo:hi],
                                           check=None if chk is None else chk[lo:hi], cum=cum, mx=mx)
        stats.append(st)
    full = None
    if chk is not None:
        full = gather_pairs(np.asarray(v, dtype=np.float64), np.arange(lo, hi), nrhs, dist, device)
    if want_cum or want_max:
        import torch
        for arr, op in ((cum, dist.ReduceOp.SUM), (mx, dist.ReduceOp.MAX)):
            if arr is None:
                continue
            t = torch.from_numpy(arr)
            t = t.to(device) if device is not None else t
            dist.all_reduce(t, op=op)
            arr[:] = t.cpu().numpy()
    return full, cum, mx, stats
