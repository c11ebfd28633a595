"""rrt-mil_amd: MI355X-native RRTEncoder forward (R-MSA + EPEG + CR-MSA).

    from rrt_mil_amd import RRTEncoder      # drop-in for modules/rrt.py::RRTEncoder
"""
import os as _os
import sys as _sys


def _claim_hw_queues():
    """HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), in creation order, and reads
    the variable ONCE, when the runtime initialises (the first HIP call -- torch does that lazily, at the first CUDA use, not
    at `import torch`).  With torch's own streams (and RCCL's) in the process, two of the four bag streams land on one queue
    and serialise: the one-stream rate, -15 %.  So the package sets 16 itself when nobody has set the
    variable and HIP is not up yet -- importing rrt_mil_amd before the first CUDA call is all a user has to do to get the
    documented multi-bag rates.  Returns what happened: 'env' (the user's setting stands), 'set' (we set 16), 'late' (HIP
    was already initialised with the default: forward_bags warns)."""
    if "GPU_MAX_HW_QUEUES" in _os.environ:
        return "env"
    t = _sys.modules.get("torch")
    if t is not None and getattr(t, "cuda", None) is not None and t.cuda.is_initialized():
        return "late"
    _os.environ["GPU_MAX_HW_QUEUES"] = "16"
    return "set"


HW_QUEUES = _claim_hw_queues()

from . import feed, geometry, sharding, synth  # noqa: F401,E402
from .feed import BagFeeder  # noqa: F401,E402
from . import _lib  # noqa: F401,E402
from ._lib import device_error  # noqa: F401,E402
from .mil import Attention, AttentionGated, DAttention, RRTMIL  # noqa: F401,E402
from .encoder import (CrossRegionAttntion, InnerAttention, RegionAttntion, RRTEncoder,  # noqa: F401
                      TransLayer, initialize_weights)

__all__ = ["RRTEncoder", "RRTMIL", "DAttention", "TransLayer", "RegionAttntion", "CrossRegionAttntion", "InnerAttention",
           "initialize_weights", "geometry", "sharding", "synth", "device_error"]
