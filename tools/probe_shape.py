#!/usr/bin/env python3
"""One-bag-in-flight forwards of one (N, region_num) shape, for rocprofv3 kernel tables of shapes off the bench configs:
    rocprofv3 --kernel-trace --stats -d /tmp/p -o p -- python tools/probe_shape.py 36000 16 [iters] [dtype]
N = 36000 at region_num = 16 is 256 regions of 144 tokens: the matrix kernels' shape of FOUR N = 9000 bags in one launch."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import RRTEncoder, geometry, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 36000
rn = int(sys.argv[2]) if len(sys.argv) > 2 else 16
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dt = sys.argv[4] if len(sys.argv) > 4 else "f32"
dev = torch.device("cuda:0")
cfg = dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=rn)
enc = RRTEncoder(**cfg).eval()
enc.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.encoder_state(**cfg).items()}, strict=True)
enc = enc.to(dev)
if dt != "f32":
    enc.compute_dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32x3": "f32x3"}[dt]
x = torch.from_numpy(synth.bag(n, 512, tag="probe")).to(dev)
y = torch.empty_like(x)
for _ in range(10):
    enc.forward_bag(x, out=y)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(iters):
    enc.forward_bag(x, out=y)
torch.cuda.synchronize()
ms = (time.perf_counter() - t) / iters * 1e3
g = geometry.region_grid(n, rn)
print(f"N={n} rn={rn} P={g.P} {dt}: {ms:.3f} ms/bag  {n / ms / 1e3:6.2f} Mtok/s", flush=True)
