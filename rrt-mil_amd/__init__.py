"""rrt-mil_amd: MI355X-native RRTEncoder forward (R-MSA + EPEG + CR-MSA).

    from rrt_mil_amd import RRTEncoder      # drop-in for modules/rrt.py::RRTEncoder
"""
from . import feed, geometry, sharding, synth  # noqa: F401
from .feed import BagFeeder  # noqa: F401
from . import _lib  # noqa: F401
from .mil import Attention, AttentionGated, DAttention, RRTMIL  # noqa: F401
from .encoder import (CrossRegionAttntion, InnerAttention, RegionAttntion, RRTEncoder,  # noqa: F401
                      TransLayer, initialize_weights)

__all__ = ["RRTEncoder", "RRTMIL", "DAttention", "TransLayer", "RegionAttntion", "CrossRegionAttntion", "InnerAttention",
           "initialize_weights", "geometry", "sharding", "synth"]
