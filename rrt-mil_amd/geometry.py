"""Region-grid geometry of the R-MSA / CR-MSA partition (host-side integers).

Restates ``RegionAttntion.padding`` / ``CrossRegionAttntion.padding``
(reference modules/rmsa.py:175-202 and :261-288, identical bodies) and the
index map of ``region_partition`` / ``region_reverse`` (modules/rmsa.py:28-54).
The C library computes the same numbers (`rrt_region_grid`, csrc/api.hip); tests pin both
against golden values taken from the reference's own ``padding()``.
"""
from dataclasses import dataclass
import math


def _ceil_sqrt(n: int) -> int:
    # int(np.ceil(np.sqrt(L))) in the reference; exact in integers for L < 2**52
    r = math.isqrt(n)
    return r if r * r == n else r + 1


@dataclass(frozen=True)
class RegionGrid:
    L: int            # tokens in the bag
    H: int            # padded grid side (H == W)
    region_size: int  # s: tokens per region side
    regions_side: int  # H // s
    add_length: int   # zero rows appended (H*H - L)

    @property
    def P(self) -> int:          # tokens per region
        return self.region_size * self.region_size

    @property
    def R(self) -> int:          # number of regions
        return self.regions_side * self.regions_side

    @property
    def Np(self) -> int:         # padded token count
        return self.H * self.H


def region_grid(L: int, region_num: int = 8, region_size: int = 0,
                min_region_num: int = 0, min_region_ratio: float = 0.0) -> RegionGrid:
    """modules/rmsa.py:175-202. ``region_size > 0`` takes precedence over ``region_num``."""
    if L <= 0:
        raise ValueError("empty bag")
    H = _ceil_sqrt(L)
    if region_size and region_size > 0:
        H += (-H) % region_size
        s = region_size
    else:
        H += (-H) % region_num
        s = H // region_num
    add = H * H - L
    # ablation escape hatch of the reference: give up region attention (one region)
    if add > L / (min_region_ratio + 1e-8) or L < min_region_num:
        H = _ceil_sqrt(L)
        H += (-H) % 2
        add = H * H - L
        s = H
    return RegionGrid(L=L, H=H, region_size=s, regions_side=H // s, add_length=add)


def token_to_slot(t: int, H: int, s: int) -> int:
    """Row of token ``t`` (padded-grid order) in the region-major [R*P] order
    produced by ``region_partition`` (modules/rmsa.py:28-39)."""
    i, j = divmod(t, H)
    rs = H // s
    return ((i // s) * rs + (j // s)) * (s * s) + (i % s) * s + (j % s)
