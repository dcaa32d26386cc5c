"""Pair-parallel multi-GPU helpers.  Pairs are independent optimisations (fresh weights per pair,
/root/reference/model/registration.py:133), so ranks never exchange data on the data path; the only
collective is the end-of-job aggregate: ONE all-reduce(SUM) of a small float64 vector (RCCL on GPUs, gloo in
tests) that carries the per-rank sums and, in one slot per rank, every rank's elapsed time -- so the maximum
over ranks (the job's wall time) and the per-rank spread come out of the same collective.

Also here: CPU / NUMA placement of a rank next to its GPU (one process per GPU, its producer thread included).
"""
import os
import time

import torch


def shard_range(n_items, rank, world):
    """Contiguous balanced shard [lo, hi) of n_items for `rank`."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def aggregate(values, elapsed, device, detail=False):
    """values: float64 vector of per-rank sums; elapsed: this rank's wall seconds.
    Returns (summed vector on CPU, max elapsed over ranks[, detail dict: world_size, elapsed_per_rank, allreduce_ms])."""
    import torch.distributed as dist
    on = dist.is_available() and dist.is_initialized()
    world, rank = (dist.get_world_size(), dist.get_rank()) if on else (1, 0)
    n = values.numel()
    v = torch.zeros(n + world, dtype=torch.float64)
    v[:n] = values.to(dtype=torch.float64, device="cpu")
    v[n + rank] = float(elapsed)                     # one slot per rank: the SUM leaves every rank's time in place (exact in float64)
    v = v.to(device)
    t0 = time.perf_counter()
    if on:
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
    v = v.cpu()                                      # (synchronises a device collective)
    dt = time.perf_counter() - t0
    per_rank = [float(x) for x in v[n:]]
    if detail:
        return v[:n], max(per_rank), {"world_size": world, "elapsed_per_rank": per_rank, "allreduce_ms": 1e3 * dt}
    return v[:n], max(per_rank)


def job_summary(pairs_local, elapsed_local, sums_local, device):
    """The bench's aggregation path: whole-job rate = (pairs of all ranks) / (slowest rank's time).
    sums_local: further float64 per-rank sums carried in the same all-reduce.  -> dict (identical on every rank)."""
    import torch.distributed as dist
    vals = torch.cat([torch.tensor([float(pairs_local)], dtype=torch.float64), torch.as_tensor(sums_local, dtype=torch.float64).reshape(-1)])
    # per-rank pair counts travel in their own one-hot slots too (ranks may hold different numbers of pairs)
    on = dist.is_available() and dist.is_initialized()
    world, rank = (dist.get_world_size(), dist.get_rank()) if on else (1, 0)
    slots = torch.zeros(world, dtype=torch.float64)
    slots[rank] = float(pairs_local)
    tot, t_max, det = aggregate(torch.cat([vals, slots]), elapsed_local, device, detail=True)
    n = vals.numel()
    pairs_rank = [float(x) for x in tot[n:]]
    rates = [p / t if t > 0 else 0.0 for p, t in zip(pairs_rank, det["elapsed_per_rank"])]
    return {"pairs": float(tot[0]), "elapsed": t_max, "value": float(tot[0]) / t_max if t_max > 0 else 0.0,
            "sums": tot[1:n], "world_size": world, "pairs_per_rank": pairs_rank, "elapsed_per_rank": det["elapsed_per_rank"],
            "rank_pairs_per_s_min": min(rates), "rank_pairs_per_s_max": max(rates), "allreduce_ms": det["allreduce_ms"]}


# ------------------------------------------------------------------------------------------------ CPU / NUMA placement
def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.extend(range(int(a), int(b or a) + 1))
    return cpus


def gpu_numa_node(index):
    """NUMA node of HIP device `index` from sysfs (PCI address of the device), or -1."""
    try:
        p = torch.cuda.get_device_properties(index)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            return int(f.read().strip())
    except Exception:
        return -1


def node_cpus(node):
    try:
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            return _parse_cpulist(f.read())
    except Exception:
        return []


def plan_affinity(local_rank, local_world, nodes, cpus_of_node, allowed):
    """CPU set of local rank `local_rank` of `local_world`: nodes[i] = NUMA node of local GPU i (-1 unknown).  Ranks whose GPUs
    sit on one node share that node's allowed CPUs in disjoint contiguous slices (a rank = main thread + pair producer + a few
    torch threads); unknown topology: a disjoint slice of everything allowed.  Never returns an empty set."""
    allowed = sorted(allowed)
    node = nodes[local_rank] if local_rank < len(nodes) else -1
    pool = [c for c in cpus_of_node(node) if c in set(allowed)] if node >= 0 else []
    if pool:
        peers = [r for r in range(local_world) if r < len(nodes) and nodes[r] == node]
    else:
        pool, peers = allowed, list(range(local_world))
    k, cnt = peers.index(local_rank), len(peers)
    lo, hi = k * len(pool) // cnt, (k + 1) * len(pool) // cnt
    return set(pool[lo:hi]) or set(pool) or set(allowed)


def pin_rank_to_gpu_numa(local_rank, local_world):
    """Restrict this process (and the threads it starts later: the pair producer) to CPUs of the NUMA node of its GPU.
    Returns a short description for the report; never raises."""
    try:
        allowed = os.sched_getaffinity(0)
        nodes = [gpu_numa_node(i) for i in range(local_world)]
        cpus = plan_affinity(local_rank, local_world, nodes, node_cpus, allowed)
        os.sched_setaffinity(0, cpus)
        return {"numa_node": nodes[local_rank], "cpus": len(cpus), "first_cpu": min(cpus)}
    except Exception as e:                       # no sched_setaffinity, no sysfs: run unpinned
        return {"numa_node": -1, "cpus": 0, "error": type(e).__name__}
