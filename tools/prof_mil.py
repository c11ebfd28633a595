#!/usr/bin/env python3
"""N RRTMIL training steps (C16-R50 config: forward + backward + Adam) for rocprofv3 --kernel-trace --stats; prints wall time per step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import RRTMIL, synth
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
fused = len(sys.argv) > 3 and sys.argv[3] == "fused"
torch.manual_seed(0)
mil = RRTMIL(input_dim=1024, n_classes=2, epeg_k=15, crmsa_k=1, all_shortcut=True, trans_dropout=0., dropout=0.25).to(dev).train()
opt = torch.optim.Adam(mil.parameters(), lr=2e-4, fused=fused)
feats = torch.from_numpy(synth.bag(N, 1024, tag="btm", nonneg=True)).to(dev).unsqueeze(0)
label = torch.zeros(1, dtype=torch.long, device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.functional.cross_entropy(mil(feats), label)
    loss.backward()
    opt.step()
for _ in range(5): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize()
print(f"RRTMIL N={N}: train step (fwd+bwd+Adam{' fused' if fused else ''}) {(time.perf_counter() - t) / steps * 1e3:.3f} ms wall")
