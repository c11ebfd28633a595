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


def _run_bench(*argv):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True,
                         timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout            # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_self_spawns_ranks_like_the_driver_invokes_it():
    """`python bench.py --gpus 2 ...` with NO torch.distributed environment (the driver's SCALE command shape): bench.py
    re-executes itself under torch.distributed.run, every rank runs the rank program (config 4: the 64-bag mix split
    by sharding.assign_bags), rank 0 prints one JSON line.  --stub-cpu swaps the GPU workload for a CPU stand-in over
    gloo; everything else (spawn, sharding, barrier, MAX over ranks, the record) is the code the GPUs run."""
    from rrt_mil_amd import sharding
    rec = _run_bench("--gpus", "2", "--stub-cpu", "--steps", "3", "--warmup", "1", "--config", "4")
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "strong"
    assert rec["unit"] == "slides/s" and rec["higher_is_better"] is True and rec["value"] > 0
    assert rec["config"]["bags_per_step"] == 64 and rec["config"]["baseline_config_index"] == 4
    sizes = [int(v) for v in np.random.RandomState(2021).randint(3000, 15001, size=64)]
    want = sharding.assign_bags(sizes, 2, region_num=8, epeg_k=21, crmsa_k=5)
    assert rec["config"]["bags_this_rank"] == len(want[0])
    assert abs(rec["value"] - 64 * 3 / (rec["ms_per_step"] * 3e-3)) / rec["value"] < 1e-3


def test_bench_single_rank_and_weak_scaling_record():
    rec1 = _run_bench("--stub-cpu", "--steps", "2", "--warmup", "0", "--config", "3")
    assert rec1["n_gpus"] == 1 and rec1["scaling"] == "weak" and rec1["config"]["n_tokens"] == 30000
    rec2 = _run_bench("--gpus", "2", "--stub-cpu", "--steps", "2", "--warmup", "0", "--config", "3")
    assert rec2["n_gpus"] == 2 and rec2["config"]["bags_per_step"] == 2 * rec1["config"]["bags_per_step"]


def test_bench_world8_configs_3_and_4_stub():
    """The driver's 8-GPU SCALE command shape, on CPU: `python bench.py --gpus 8 --config c` with no torch.distributed
    environment.  Eight ranks over gloo (the stand-in for RCCL), one JSON line with n_gpus == 8; configs[3] = every rank its
    own N=30000 bags ("weak"), configs[4] = the 64-bag mix LPT-split eight ways with <= 5 % cost imbalance ("strong").  The
    only collectives in the rank program are a barrier and one MAX all-reduce of the elapsed time -- both exist in RCCL."""
    from rrt_mil_amd import sharding
    rec = _run_bench("--gpus", "8", "--stub-cpu", "--steps", "2", "--warmup", "1", "--config", "4")
    assert rec["n_gpus"] == 8 and rec["scaling"] == "strong" and rec["data"] == "stub"
    assert rec["config"]["bags_per_step"] == 64 and rec["config"]["collectives"]["world"] == 8
    assert rec["config"]["collectives"]["backend"] == "gloo"
    assert "barrier + MAX all-reduce" in rec["config"]["collectives"]["use"]
    sizes = [int(v) for v in np.random.RandomState(2021).randint(3000, 15001, size=64)]
    kw = dict(region_num=8, epeg_k=21, crmsa_k=5)
    a = sharding.assign_bags(sizes, 8, **kw)
    assert sorted(i for r in a for i in r) == list(range(64))
    loads = [sum(sharding.bag_cost(sizes[i], **kw) for i in r) for r in a]
    assert max(loads) / (sum(loads) / 8) <= 1.05, "LPT imbalance of the configs[4] mix over 8 ranks"
    assert rec["config"]["bags_this_rank"] == len(a[0])
    assert abs(rec["value"] - 64 * 2 / (rec["ms_per_step"] * 2e-3)) / rec["value"] < 1e-3
    rec3 = _run_bench("--gpus", "8", "--stub-cpu", "--steps", "2", "--warmup", "0", "--config", "3")
    assert rec3["n_gpus"] == 8 and rec3["scaling"] == "weak" and rec3["config"]["n_tokens"] == 30000
    one = _run_bench("--stub-cpu", "--steps", "2", "--warmup", "0", "--config", "3")
    assert rec3["config"]["bags_per_step"] == 8 * one["config"]["bags_per_step"]
    # configs[3] on 8 ranks as the BASELINE words it (8 bags, one per GPU): exactly one each, imbalance 1.0
    assert sharding.assign_bags([30000] * 8, 8, region_num=16) == [[i] for i in range(8)]


def test_rank_program_uses_only_rccl_shaped_collectives():
    """No data-path collective: the only torch.distributed calls in bench.py and the package are init / barrier /
    all_reduce / get_backend / get_world_size / destroy -- all of which RCCL implements (no gloo-only object collectives, no
    send / recv of activations)."""
    import re
    allowed = {"init_process_group", "barrier", "all_reduce", "get_backend", "get_world_size", "destroy_process_group",
               "is_available", "is_initialized", "ReduceOp", "get_rank"}
    for rel in ("bench.py", os.path.join("rrt-mil_amd", "sharding.py")):
        src = open(os.path.join(ROOT, rel)).read()
        used = set(re.findall(r"\bdist\.([A-Za-z_]+)", src))
        assert used <= allowed, (rel, sorted(used - allowed))
