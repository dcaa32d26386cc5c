"""Pair-parallel multi-GPU helpers.  Pairs are independent optimisations (fresh weights per pair,
/root/reference/model/registration.py:133), so ranks never exchange data on the data path; the only
collective is the end-of-job aggregate (one SUM + one MAX all-reduce, RCCL on GPUs, gloo in tests)."""
import torch


def shard_range(n_items, rank, world):
    """Contiguous balanced shard [lo, hi) of n_items for `rank`."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def aggregate(values, elapsed, device):
    """values: float64 vector of per-rank sums; elapsed: this rank's wall seconds.
    Returns (summed vector on CPU, max elapsed over ranks)."""
    import torch.distributed as dist
    v = values.to(device=device, dtype=torch.float64).clone()
    t = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return v.cpu(), float(t.item())
