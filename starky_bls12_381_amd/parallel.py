"""Proof-parallel multi-GPU helpers: one process per GPU, torch.distributed (nccl = RCCL on ROCm, gloo on CPU).

The starky hot path shards at PROOF granularity: the six proofs of one signature verification are independent once
the natives are known (reference src/aggregate_proof.rs:304-370), so there is no collective on the data path -- only
job assignment, an optional broadcast of the (small) public inputs from the rank that parsed the input, a barrier and
a max-reduce of the wall time.
"""
import os

import numpy as np

# what one proof of each AIR costs a pool: ms per proof with the pool full of that AIR on one MI355X (tools/air_pool_cost.py,
# profiles/r06_air_pool_cost.json; the same table as csrc/scheduler.cpp air_cost -- FinalExp dominates, as on the reference's CPU, README.md:36-39)
AIR_COST = {3: 128.0, 2: 24.0, 1: 10.8, 0: 15.4, 4: 12.6}


def rank_info():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_distributed(backend=None):
    """Join the process group described by the torchrun environment; returns torch.distributed or None (single process)."""
    rank, local_rank, world = rank_info()
    if world <= 1:
        return None
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, **kwargs)
    return dist


def assign_jobs(costs, world_size):
    """Longest-processing-time-first assignment of independent proof jobs to ranks.
    costs: list of floats; returns per-rank lists of job indices (every job exactly once)."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * world_size
    out = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += costs[i]
    return out


def broadcast_u64(dist, arr, src=0, device="cpu"):
    """Broadcast a uint64 numpy array (e.g. the packed public inputs, <= 5064 words per proof) from rank `src`."""
    import torch
    a = np.ascontiguousarray(arr, dtype=np.uint64)
    if dist is None:
        return a
    t = torch.from_numpy(a.view(np.int64).copy()).to(device)
    dist.broadcast(t, src=src)
    return t.cpu().numpy().view(np.uint64)


def max_over_ranks(dist, value, device="cpu"):
    """Wall time of the slowest rank (the benchmark's `elapsed`)."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(dist, values, device="cpu"):
    """Every rank's list of floats, on every rank: [[rank 0's values], [rank 1's], ...] (one all_gather of a small tensor)."""
    vals = [float(v) for v in values]
    if dist is None:
        return [vals]
    import torch
    t = torch.tensor(vals, dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(x) for x in o.cpu()] for o in out]


def sum_over_ranks(dist, value, device="cpu"):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
