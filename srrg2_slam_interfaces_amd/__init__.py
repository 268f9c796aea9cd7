"""MI355X-native multi-cue aligner hot path of srrg2_slam_interfaces."""
from . import _abi as abi  # noqa: F401
