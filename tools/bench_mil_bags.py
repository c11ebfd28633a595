#!/usr/bin/env python3
"""Slides/s of the whole classifier (RRTMIL, C16-R50 shape: 1024 -> 512 fc + ReLU, encoder, DAttention, predictor) through
RRTMIL.forward_bags: N = 9000 x 1024 slides resident in HBM, 1 .. 4 slides in flight, fp32 and bf16.
    python tools/bench_mil_bags.py"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import RRTMIL, synth   # noqa: E402

dev = torch.device("cuda:0")
mil = RRTMIL(input_dim=1024, n_classes=2, epeg_k=15, crmsa_k=1, all_shortcut=True).eval()
mil.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.mil_state(input_dim=1024, n_classes=2, epeg_k=15, crmsa_k=1).items()})
mil = mil.to(dev)
base = [torch.from_numpy(synth.bag(9000, 1024, tag=f"mb/{i}", nonneg=True)).to(dev) for i in range(4)]
bags = [base[i % 4] for i in range(128)]
for dt in (None, torch.bfloat16):
    mil.online_encoder.compute_dtype = dt
    for S in (1, 2, 3, 4):
        mil.forward_bags(bags[:16], streams=S)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            mil.forward_bags(bags, streams=S)
        torch.cuda.synchronize()
        print(f"RRTMIL C16 N=9000 {'fp32' if dt is None else 'bf16'} forward_bags(128 slides, streams={S}): {3 * len(bags) / (time.perf_counter() - t):.0f} slides/s")
