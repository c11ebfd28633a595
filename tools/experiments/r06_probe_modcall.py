#!/usr/bin/env python3
"""Round 6 probe: why does bench.py::module_call's asynchronous forward_bags variant (caller under its own stream) read 4.4-4.5 k
slides/s fp32 when the timed region -- the same call -- reads 5.27 k?  One process per ORDER:
    python tools/experiments/r06_probe_modcall.py <dtype> <order: comma list of default|side> [loop_first: 0|1]"""
import contextlib, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rrt_mil_amd import RRTEncoder, synth
dt, order = sys.argv[1], sys.argv[2].split(",")
loop_first = len(sys.argv) > 3 and sys.argv[3] == "1"
dev = torch.device("cuda:0")
cfg = dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8)
enc = RRTEncoder(**cfg).eval()
enc.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.encoder_state(**cfg).items()}, strict=True)
enc = enc.to(dev)
enc.compute_dtype = {"f32": torch.float32, "bf16": torch.bfloat16}[dt]
caller = torch.cuda.Stream(dev)
bags = [torch.from_numpy(synth.bag(9000, 512, tag=f"pm/{i}")).to(dev).unsqueeze(0) for i in range(4)]
batch = [bags[i % 4] for i in range(256)]
outs = [torch.empty_like(b[0]) for b in batch]
res = []
with torch.no_grad():
    if loop_first:
        for i in range(264):
            y = enc(bags[i % 4])
        torch.cuda.synchronize()
    for name in order:
        ctx = contextlib.nullcontext() if name == "default" else torch.cuda.stream(caller)
        with ctx:
            for _ in range(2):
                enc.forward_bags(batch, streams=4, outs=outs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                enc.forward_bags(batch, streams=4, outs=outs)
            torch.cuda.synchronize()
            res.append((name, round(5 * 256 / (time.perf_counter() - t0), 1)))
print(dt, "loop_first" if loop_first else "", res)
