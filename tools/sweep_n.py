#!/usr/bin/env python3
"""ms per bag of the encoder forward (fp32, one bag in flight) over bag sizes; run once plain and once with
RRT_NO_FUSED=1 -- on a -DRRT_TUNING library (tools/build_ablation.sh tune -DRRT_TUNING; RRT_HIP_LIB=tools/_abl/librrt_tune.so:
the product library does not read the environment) -- to compare the fused R-MSA kernel with the unfused linear + attention
pair."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import RRTEncoder, geometry, synth  # noqa: E402

dev = torch.device("cuda:0")
cfg = dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8)
enc = RRTEncoder(**cfg).eval()
enc.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.encoder_state(**cfg).items()}, strict=True)
enc = enc.to(dev)
if os.environ.get("SWEEP_DTYPE"):                   # "bf16" / "f16" / "f32x3": the reduced modes' one-bag times
    enc.compute_dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32x3": "f32x3"}[os.environ["SWEEP_DTYPE"]]
big = torch.from_numpy(synth.bag(16000, 512, tag="sweep")).to(dev)
for n in [int(a) for a in sys.argv[1:]] or [2500, 3000, 4096, 5000, 6000, 7000, 8000, 9000, 10500, 12000, 15000]:
    x = big[:n].contiguous()
    y = torch.empty_like(x)
    for _ in range(10):
        enc.forward_bag(x, out=y)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(100):
        enc.forward_bag(x, out=y)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 100 * 1e3
    g = geometry.region_grid(n, 8)
    print(f"N={n:6d} P={g.P:4d}: {ms:.3f} ms/bag  {1e3 / ms:7.0f} slides/s  {n / ms / 1e3:6.2f} Mtok/s", flush=True)
