"""Round-4 fixtures from the REAL reference at batch > 1 (run in the build container; arrays only):

    python tools/make_golden_batch.py

G20: RRTEncoder.forward on (B, N, D) input with B = 2 / 3 (modules/rrt.py:165-202).  At B > 1 the reference's
region_partition puts the regions of all bags into one leading axis (modules/rmsa.py:28-39), so CR-MSA's inner attention
runs over the 64 B representatives of ALL bags (rmsa.py:316-322): the bags are coupled there and nowhere else.  Bag b
is synth.bag(N, 512, tag=f"batch/b{b}"); sampled output rows per bag + checksums per bag.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_golden import STATE_KEYS, checksums, save, cfg_array  # noqa: E402
from rrt_mil_amd import synth  # noqa: E402
from _ref import build_reference_encoder  # noqa: E402


def main():
    cases = [("default", dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8), 2, 1500),
             ("default", dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8), 3, 700),
             ("c16", dict(mlp_dim=512, epeg_k=15, crmsa_k=1, region_num=8, all_shortcut=True), 2, 1000),
             ("nsclc", dict(mlp_dim=512, epeg_k=21, crmsa_k=5, region_num=8), 2, 3100),
             ("mlp", dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8, crmsa_mlp=True), 2, 900),
             ("heads1", dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8, crmsa_heads=1), 2, 800),
             ("crmsa_only", dict(mlp_dim=512, crmsa_k=3, n_layers=1), 2, 1200)]
    for name, cfg, B, N in cases:
        state = synth.encoder_state(**{k: v for k, v in cfg.items() if k in STATE_KEYS})
        enc = build_reference_encoder(state, **cfg)
        x = np.stack([synth.bag(N, 512, tag=f"batch/b{b}") for b in range(B)])
        with torch.no_grad():
            y = enc(torch.from_numpy(x)).numpy()
            y1 = enc(torch.from_numpy(x[:1])).numpy()            # bag 0 alone: differs from y[0] (the coupling is real)
        rows = np.unique(np.concatenate([np.arange(0, N, max(1, N // 60)), [N - 1]]))
        save(f"G20_batch_{name}_b{B}_n{N}", cfg=cfg_array(cfg), n=np.array(N), b=np.array(B), rows=rows,
             y_rows=y[:, rows], y_sums=np.stack([checksums(y[b]) for b in range(B)]),
             coupling=np.array(float(np.abs(y[0] - y1[0]).max())))


if __name__ == "__main__":
    main()
