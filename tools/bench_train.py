#!/usr/bin/env python3
"""Training step time (forward with stash + backward, fp32, one bag per step) of the encoder and of the whole
RRTMIL classifier (cross-entropy on the bag label), drop_out = 0.
    python tools/bench_train.py [N]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import RRTEncoder, RRTMIL, synth
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 9000

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

enc = RRTEncoder(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8, drop_out=0.).to(dev).train()
x = torch.from_numpy(synth.bag(N, 512, tag="bt")).to(dev).unsqueeze(0)
G = torch.randn(1, N, 512, device=dev)
def enc_step():
    enc.zero_grad(set_to_none=True)
    y = enc(x)
    (y * G).sum().backward()
def enc_fwd():
    with torch.no_grad(): enc(x)
print(f"encoder N={N}: train step (fwd+bwd) {timeit(enc_step):.3f} ms   inference forward {timeit(enc_fwd):.3f} ms")
def enc_step_amp():
    enc.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = enc(x)
    (y.float() * G).sum().backward()
print(f"  under bf16 autocast: train step {timeit(enc_step_amp):.3f} ms")
mem0 = torch.cuda.max_memory_allocated() / 1e6
print(f"  peak device memory so far {mem0:.0f} MB")

feats = torch.from_numpy(synth.bag(N, 1024, tag="btm", nonneg=True)).to(dev).unsqueeze(0)
# The step launches ~240 kernels, 160 of them the seven element-wise kernels per parameter of torch's default (foreach-less)
# Adam: with that optimizer the step is bound by the host's launch rate (1.7-2.5 ms from box to box, GPU kernel time 1.85 ms
# under rocprofv3), so the fused optimizer is timed beside it.
for fused in (False, True):
    torch.manual_seed(0)
    mil = RRTMIL(input_dim=1024, n_classes=2, epeg_k=15, crmsa_k=1, all_shortcut=True, trans_dropout=0., dropout=0.25).to(dev).train()
    opt = torch.optim.Adam(mil.parameters(), lr=2e-4, fused=fused)
    with torch.no_grad():
        label = mil.eval()(feats).argmin(-1)      # the class the fresh model likes least: a loss worth descending
    mil.train()
    def mil_step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(mil(feats), label)
        loss.backward()
        opt.step()
        return loss
    l0 = mil_step().item()
    print(f"RRTMIL (C16-R50 config) N={N}: train step (fwd+bwd+Adam{', fused=True' if fused else ''}) {timeit(mil_step):.3f} ms")
    print(f"  loss {l0:.4f} at the first step -> {mil_step().item():.6f} after 36 Adam steps on the same bag")
