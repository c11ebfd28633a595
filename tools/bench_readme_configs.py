#!/usr/bin/env python3
"""ms/slide of the six published README configs (whole RRTMIL classifier, fp32, one bag in flight) at N=9000."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrt_mil_amd import RRTMIL, synth
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
readme = {
    "c16_r50": dict(input_dim=1024, epeg_k=15, crmsa_k=1, all_shortcut=True),
    "c16_plip": dict(input_dim=512, epeg_k=9, crmsa_k=3, all_shortcut=True),
    "brca_r50": dict(input_dim=512, epeg_k=17, crmsa_k=3, crmsa_heads=1),
    "brca_plip": dict(input_dim=512, crmsa_k=1, all_shortcut=True),
    "nsclc_r50": dict(input_dim=512, epeg_k=21, crmsa_k=5),
    "nsclc_plip": dict(input_dim=512, epeg_k=13, crmsa_k=3, crmsa_heads=1, all_shortcut=True, crmsa_mlp=True),
}
ONLY = os.environ.get("README_ONLY")
for tag, extra in readme.items():
    if ONLY and tag != ONLY:
        continue
    cfg = dict(n_classes=2, da_act="tanh", act="relu"); cfg.update(extra)
    mil = RRTMIL(**cfg).eval().to(dev)
    x = torch.from_numpy(synth.bag(N, cfg["input_dim"], tag="rb", nonneg=True)).to(dev).unsqueeze(0)
    for _ in range(10): mil(x)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(100): mil(x)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t) * 10
    print(f"{tag:11s} {ms:.3f} ms/slide  {1e3 / ms:6.0f} slides/s")
