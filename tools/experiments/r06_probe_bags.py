#!/usr/bin/env python3
"""Round 6 probe: where does forward_bags lose against the raw C-ABI loop?  One measurement per process (stream -> queue map).
    python tools/experiments/r06_probe_bags.py <dtype f32|bf16> <nb bags/call> <inputs: cyc|fresh> <outs: alias|fresh> <caller: default|side>
prints slides/s.  `cyc` = 4 distinct inputs cycled (bench.py's timed region), `fresh` = 28 distinct; `alias` = 8 output buffers
reused, `fresh` = one per bag of the call."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rrt_mil_amd import RRTEncoder, synth

dtype, nb, inp, outm, caller = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
dev = torch.device("cuda:0")
cfg = dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8)
enc = RRTEncoder(**cfg).eval()
enc.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.encoder_state(**{k: v for k, v in cfg.items() if k != "region_num"}).items()}, strict=True)
enc = enc.to(dev)
if dtype == "bf16":
    enc.compute_dtype = torch.bfloat16
nd = 4 if inp == "cyc" else 28
base = [torch.randn(9000, 512, device=dev) for _ in range(nd)]
bags = [base[i % nd] for i in range(nb)]
if outm == "alias":
    ob = [torch.empty(9000, 512, device=dev) for _ in range(8)]
    outs = [ob[i % 8] for i in range(nb)]
else:
    outs = [torch.empty(9000, 512, device=dev) for _ in range(nb)]
    for o in outs:
        o.zero_()
if caller == "side":
    torch.cuda.set_stream(torch.cuda.Stream(dev))
with torch.no_grad():
    for _ in range(max(2, 200 // nb)):
        enc.forward_bags(bags, streams=4, outs=outs)
    torch.cuda.synchronize()
    reps = max(3, 1500 // nb)
    t = time.perf_counter()
    for _ in range(reps):
        enc.forward_bags(bags, streams=4, outs=outs)
    host = time.perf_counter() - t
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
print(f"{dtype} nb={nb} in={inp} out={outm} caller={caller}: {nb * reps / dt:.0f} slides/s  (host {host / (nb * reps) * 1e6:.1f} us/bag)")
