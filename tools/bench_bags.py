#!/usr/bin/env python3
"""Batch-of-bags throughput through the executor (rrt_executor_forward), fp32.
    python tools/bench_bags.py                 # table: uniform N=9000 batches and the BASELINE configs[4] mix
    python tools/bench_bags.py uniform NB S    # one measurement (what the table spawns, one process each)
    python tools/bench_bags.py mix S
configs[4]: epeg_k=21, crmsa_k=5, D=512, N_i ~ randint(3000, 15001) (seed 2021), 64 bags, independent B=1 forwards.
Every measurement runs in a fresh process: HIP maps streams to hardware queues in creation order, and a
process that creates and destroys executors with different stream counts ends up with colliding queues
(measured: 2.2 k instead of 3.6 k slides/s at S=1 after such churn) -- applications create ONE executor."""
import os
import subprocess
import sys
import time

# HWQ=unset: leave GPU_MAX_HW_QUEUES to the package (rrt_mil_amd sets 16 at import when HIP is not up yet); HWQ=<n>: force n;
# HWQ=torch_first: `import torch` + a CUDA call BEFORE the package is imported (the "late" case: HIP's default 4 holds)
_hwq = os.environ.get("HWQ", "unset")
if _hwq == "torch_first":
    import torch as _t
    _t.zeros(1, device="cuda:0")
elif _hwq != "unset":
    os.environ["GPU_MAX_HW_QUEUES"] = _hwq


def measure(kind, nb, streams):
    import numpy as np
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from rrt_mil_amd import RRTEncoder, synth
    dev = torch.device("cuda:0")
    cfg = (dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8) if kind == "uniform"
           else dict(mlp_dim=512, epeg_k=21, crmsa_k=5, region_num=8))
    enc = RRTEncoder(**cfg).eval()
    enc.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.encoder_state(**cfg).items()}, strict=True)
    enc = enc.to(dev)
    if os.environ.get("BAGS_DTYPE") == "bf16":        # the --amp path (configs[4] is quoted in bf16)
        enc.compute_dtype = torch.bfloat16
    if kind == "uniform":
        base = [torch.from_numpy(synth.bag(9000, 512, tag=f"bb/{i}")).to(dev) for i in range(4)]
        bags = [base[i % 4] for i in range(nb)]
    else:
        sizes = np.random.RandomState(2021).randint(3000, 15001, size=64)
        big = torch.from_numpy(synth.bag(15000, 512, tag="bb/mix")).to(dev)
        bags = [big[:int(n)].contiguous() for n in sizes]
    outs = [torch.empty_like(b) for b in bags]
    if os.environ.get("BAGS_ALIAS_OUTS"):         # (experiment: the outputs of all bags of a stream land in few buffers)
        k = int(os.environ["BAGS_ALIAS_OUTS"])
        outs = [outs[i % k] if bags[i].shape == bags[i % k].shape else outs[i] for i in range(len(bags))]
    reps = max(3, 600 // len(bags))               # >= ~150 ms of GPU work
    if os.environ.get("BAGS_CALLER_STREAM"):      # (experiment: the caller's stream is not the process's default stream)
        torch.cuda.set_stream(torch.cuda.Stream(dev))
    for _ in range(max(1, 100 // len(bags))):     # warm: executor creation, clocks
        enc.forward_bags(bags, streams=streams, outs=outs)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        enc.forward_bags(bags, streams=streams, outs=outs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    tokens = sum(b.size(0) for b in bags)
    print(f"{len(bags) * reps / dt:.0f} {tokens * reps / dt / 1e6:.2f}")


def spawn(*a):
    out = subprocess.run([sys.executable, os.path.abspath(__file__)] + [str(x) for x in a], capture_output=True,
                         text=True, timeout=600)
    if out.returncode:
        raise SystemExit(out.stderr[-2000:])
    return out.stdout.strip().split()[-2:]


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "uniform":
        measure("uniform", int(sys.argv[2]), int(sys.argv[3]))
    elif len(sys.argv) > 1 and sys.argv[1] == "mix":
        measure("mix", 64, int(sys.argv[2]))
    elif len(sys.argv) > 1 and sys.argv[1] == "hwq":
        # what a user gets at 64 bags / call, S = 4, depending on who set GPU_MAX_HW_QUEUES (the child inherits HWQ)
        for mode in ("unset", "4", "16", "torch_first"):
            os.environ["HWQ"] = mode
            print(f"HWQ={mode:12s} uniform N=9000, 64 bags/call, S=4: {spawn('uniform', 64, 4)[0]} slides/s", flush=True)
    else:
        for nb in (2, 4, 8, 16, 64):
            print(f"uniform N=9000, {nb:3d} bags/call:", "  ".join(f"S={s}: {spawn('uniform', nb, s)[0]:>5s}/s" for s in (1, 2, 3, 4)),
                  flush=True)
        for s in (1, 2, 3, 4):
            r = spawn("mix", s)
            print(f"configs[4] mix (64 bags/call, N in [3000,15000], epeg_k=21 crmsa_k=5) S={s}: {r[0]} slides/s  {r[1]} Mtokens/s",
                  flush=True)
