"""Multi-GPU layer: independent QPs are sharded across ranks, one process per GPU.

The path has NO exchange step: every QP is solved entirely on one GPU, so
the only collectives are the ones a driver needs around the solve -- a barrier, the MAX of the
elapsed time and a tiny all-gather of per-rank counters.  backend "nccl" is RCCL on ROCm; the
CPU tests run the same code over "gloo".
"""
import numpy as np

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None


def shard_indices(N, rank, world):
    """Interleaved assignment QP k -> rank k mod world: iteration-count variance averages out
    across GPUs better than with contiguous blocks."""
    return np.arange(rank, N, world, dtype=np.int64)


def is_distributed():
    return dist is not None and dist.is_available() and dist.is_initialized()


def max_over_ranks(value, device="cpu"):
    """MAX all-reduce of a python float (the bench contract's elapsed time)."""
    if not is_distributed():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)   # (also with one rank: the same collective at every size)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_counters(counters, device="cpu"):
    """All-gather a fixed-length list of per-rank counters (e.g. [n_solved, sum_iter, elapsed]).
    Returns an array (world, len(counters)) on every rank."""
    c = np.asarray(counters, dtype=np.float64)
    if not is_distributed() or dist.get_world_size() == 1:
        return c[None, :]
    t = torch.tensor(c, dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out).cpu().numpy()


def solve_sharded(solve_fn, batch, rank=None, world=None, device="cpu"):
    """Solve this rank's shard of a host-resident batch (dict of arrays with leading dimension N) with
    `solve_fn(shard) -> dict(x, lam, exitflag, iter)` and return (indices, result, counters[world,3]).
    The product passes daqp_amd.solve_batch; the CPU tests pass a stand-in."""
    import time
    if rank is None:
        rank = dist.get_rank() if is_distributed() else 0
    if world is None:
        world = dist.get_world_size() if is_distributed() else 1
    N = batch["f"].shape[0]
    idx = shard_indices(N, rank, world)
    shard = {k: (v[idx] if hasattr(v, "shape") and v.shape[:1] == (N,) else v) for k, v in batch.items()}
    if is_distributed():
        dist.barrier()
    t0 = time.perf_counter()
    res = solve_fn(shard)
    dt = time.perf_counter() - t0
    counters = gather_counters([idx.size, float(np.sum(res["iter"])), dt], device=device)
    return idx, res, counters
