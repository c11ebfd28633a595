"""rrt-mil_amd: MI355X-native RRTEncoder forward (R-MSA + EPEG + CR-MSA)."""
from . import geometry, synth  # noqa: F401
