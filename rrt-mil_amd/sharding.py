"""Bag-parallel sharding across the GPUs of one node.

One WSI bag is one independent unit of work (and must stay one: at batch > 1 the
reference couples bags inside CR-MSA).  Ranks never exchange activations; the only
collectives are a barrier and scalar reductions of counters / elapsed time, which is
all `torch.distributed` (RCCL on the GPUs, gloo in the CPU tests) is used for.
"""
from typing import List, Sequence

from .geometry import region_grid


def bag_cost(n_tokens: int, dim: int = 512, region_num: int = 8, heads: int = 8, epeg_k: int = 15,
             crmsa_k: int = 3) -> float:
    """Algorithmic FLOPs of one forward -- the balancing weight."""
    g, g8 = region_grid(n_tokens, region_num), region_grid(n_tokens, 8)
    D, k = dim, crmsa_k
    return float(8 * g.Np * D * D + 4 * g.Np * g.P * D + 2 * g.Np * g.P * heads * epeg_k
                 + 6 * g8.Np * D * k + 8 * k * 64 * D * D + 4 * k * 64 * 64 * D)


def assign_bags(sizes: Sequence[int], world_size: int, **cost_kw) -> List[List[int]]:
    """Greedy longest-processing-time split: returns, per rank, the indices of its bags.
    Deterministic (ties by index), every bag assigned exactly once."""
    loads = [0.0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    order = sorted(range(len(sizes)), key=lambda i: (-bag_cost(sizes[i], **cost_kw), i))
    for i in order:
        r = min(range(world_size), key=lambda q: (loads[q], q))
        out[r].append(i)
        loads[r] += bag_cost(sizes[i], **cost_kw)
    for r in range(world_size):
        out[r].sort()
    return out


def max_over_ranks(value: float, device=None) -> float:
    """MAX-reduce a python float over the default process group (no-op without one).  With a group of ONE rank the
    collective still runs: `torch.distributed.run --nproc-per-node 1` then exercises RCCL end to end (init, a device
    all-reduce, teardown) on one GPU -- what tests/test_multigpu_plumbing.py relies on."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
