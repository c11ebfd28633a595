#!/usr/bin/env python3
"""N encoder training steps (forward with stash + backward, fp32, drop_out = 0.1) for rocprofv3 --kernel-trace --stats."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import RRTEncoder, synth
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
enc = RRTEncoder(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8, drop_out=0.1).to(dev).train()
x = torch.from_numpy(synth.bag(N, 512, tag="bt")).to(dev).unsqueeze(0)
G = torch.randn(1, N, 512, device=dev)
def step():
    enc.zero_grad(set_to_none=True)
    (enc(x) * G).sum().backward()
for _ in range(5): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize()
print(f"encoder N={N}: train step (fwd+bwd) {(time.perf_counter() - t) / steps * 1e3:.3f} ms")
