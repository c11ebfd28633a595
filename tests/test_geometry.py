"""CPU: host-side region geometry against the reference's own padding() output
(golden G0, modules/rmsa.py:175-202) and the partition index map (:28-54)."""
import numpy as np

from conftest import load_golden
from rrt_mil_amd.geometry import region_grid, token_to_slot
from oracle import rrt_oracle as O


def test_grid_matches_reference_padding():
    tab = load_golden("G0_geometry")["table"]
    for L, rn, rs, mrn, mrr, H, rsz, add in tab:
        g = region_grid(int(L), int(rn), int(rs), int(mrn), float(mrr))
        assert (g.H, g.region_size, g.add_length) == (int(H), int(rsz), int(add)), (L, rn, rs, mrn, mrr)
        assert O.grid(int(L), int(rn), int(rs), int(mrn), float(mrr)) == (int(H), int(rsz), int(add))


def test_survey_table():
    # SURVEY.md §3.3 geometry table
    for N, rn, H, s, P, R, pad in ((512, 8, 24, 3, 9, 64, 64), (3000, 8, 56, 7, 49, 64, 136),
                                   (9000, 8, 96, 12, 144, 64, 216), (15000, 8, 128, 16, 256, 64, 1384),
                                   (30000, 16, 176, 11, 121, 256, 976), (50, 8, 8, 1, 1, 64, 14)):
        g = region_grid(N, rn)
        assert (g.H, g.region_size, g.P, g.R, g.add_length) == (H, s, P, R, pad)


def test_token_to_slot_is_partition_inverse():
    for H, s in ((24, 3), (96, 12), (8, 1), (20, 10), (6, 6)):
        perm = O.partition_index(H, s)          # slot -> token
        for t in range(0, H * H, max(1, H * H // 97)):
            assert perm[token_to_slot(t, H, s)] == t
