"""CPU, world_size 2 over gloo: the N>1 path of bench.py / INTEGRATION §4 -- cost-balanced
bag assignment, barrier, MAX/SUM reductions -- is correct by construction (no GPU)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, sizes, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from rrt_mil_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sharding.assign_bags(sizes, world)[rank]
    # "process" the bags: the checksum of every bag must be counted exactly once over all ranks
    local = float(sum(sizes[i] * (i + 1) for i in mine))
    dist.barrier()
    total = sharding.sum_over_ranks(local)
    tmax = sharding.max_over_ranks(0.1 * (rank + 1))
    nbags = sharding.sum_over_ranks(float(len(mine)))
    q.put((rank, mine, total, tmax, nbags))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bag_parallel():
    rng = np.random.RandomState(2021)
    sizes = [int(v) for v in rng.randint(3000, 15001, size=13)]     # BASELINE configs[4]: mixed N
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, sizes, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    all_idx = sorted(i for _, mine, *_ in res for i in mine)
    assert all_idx == list(range(len(sizes)))                        # each bag exactly once
    want_total = float(sum(n * (i + 1) for i, n in enumerate(sizes)))
    for _, _, total, tmax, nbags in res:
        assert total == want_total and abs(tmax - 0.2) < 1e-12 and nbags == len(sizes)


def test_assignment_is_balanced_and_deterministic():
    from rrt_mil_amd import sharding
    rng = np.random.RandomState(7)
    sizes = [int(v) for v in rng.randint(3000, 15001, size=64)]
    a = sharding.assign_bags(sizes, 8)
    assert a == sharding.assign_bags(sizes, 8)
    loads = [sum(sharding.bag_cost(sizes[i]) for i in r) for r in a]
    assert max(loads) / (sum(loads) / 8) < 1.08                      # LPT: within a few % of even
    # equal bags (BASELINE configs[3]: 8 x N=30000 on 8 GPUs) -> exactly one each
    assert sharding.assign_bags([30000] * 8, 8) == [[i] for i in range(8)]
    from oracle import rrt_oracle
    assert sharding.bag_cost(9000) == rrt_oracle.flops_per_bag(9000)
