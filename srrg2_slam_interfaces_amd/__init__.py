"""MI355X-native multi-cue aligner hot path of srrg2_slam_interfaces."""
from . import _abi as abi  # noqa: F401


def MultiAligner(variable_kind=abi.SE3_QUAT_RIGHT, device=0):
    """Product aligner on the HIP library (raises if the library or a HIP device is missing)."""
    from . import _capi
    from .aligner import MultiAligner as _MA

    return _MA(_capi.backend(), variable_kind, device)


def PoseGraph(variable_kind=abi.SE3_QUAT_RIGHT, device=0):
    """Product pose-graph solver on the HIP library (raises if the library or a HIP device is missing)."""
    from . import _capi
    from .posegraph import PoseGraph as _PG

    l = _capi.lib()
    return _PG(l, "srrg2_posegraph_", l.srrg2_amd_last_error, variable_kind, device=device)


def scene_binding(device=0):
    """Binding of the scene/clipper/merger classes of ``mapping`` to the HIP library."""
    from . import _capi
    from .mapping import _Binding

    l = _capi.lib()
    return _Binding(l, "srrg2_scene_", l.srrg2_amd_last_error, device)
