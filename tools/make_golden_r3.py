"""Round-3 fixtures from the REAL reference (run in the build container; arrays only):

    python tools/make_golden_r3.py

G18: default encoder (D=512, epeg_k=15, crmsa_k=3, region_num=8) at the bag sizes whose regions take the row-tile
counts the earlier end-to-end fixtures missed -- N = 5000 / 6000 / 7000 / 10500 / 12000 (P = 81 / 100 / 121 / 169 / 196:
MT = 6, 7, 8, 11, 13 of the fused fp32 kernel) and N = 13000 (P = 225: the hand-over to the unfused / streaming path) --
with sampled output rows, checksums and the x1 / x2 stage rows (modules/rmsa.py:183-190 geometry, rrt.py:165-202).
G19: BASELINE configs[4]'s hyper-parameters (epeg_k=21, crmsa_k=5) at sizes between the two G5 fixtures, so that the
executor's mixed-size run is pinned bag by bag (N = 5600: P = 100; N = 11000: P = 169).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_golden import run_ref, stage_rows, checksums, save, cfg_array  # noqa: E402


def main():
    cfg = dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8)
    for N in (5000, 6000, 7000, 10500, 12000, 13000):
        x, y, _, enc = run_ref(N, cfg)
        rows = np.unique(np.concatenate([np.arange(0, N, N // 120), [N - 1, N - 2]]))
        st = stage_rows(enc, x, rows)
        save(f"G18_d512_n{N}", cfg=cfg_array(cfg), n=np.array(N), rows=rows, y_rows=y[rows], y_sums=checksums(y),
             x1_rows=st["x1"], x2_rows=st["x2"])
    cfg = dict(mlp_dim=512, epeg_k=21, crmsa_k=5, region_num=8)
    for N in (5600, 11000):
        x, y, _, enc = run_ref(N, cfg)
        rows = np.arange(0, N, N // 100)
        save(f"G19_d512_n{N}_k21_c5", cfg=cfg_array(cfg), n=np.array(N), rows=rows, y_rows=y[rows], y_sums=checksums(y))


if __name__ == "__main__":
    main()
