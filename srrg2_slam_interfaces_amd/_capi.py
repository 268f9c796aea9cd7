"""ctypes loader of the product library (lib/libsrrg2_slam_amd.so = HIP kernels + C ABI).

There is NO fallback: if the shared library is missing or cannot be loaded this raises, and
creating an aligner without a HIP device fails with SRRG2_E_NO_DEVICE.  The CPU oracle under
oracle/ is test infrastructure and is never imported from here.
"""
import ctypes as C
import os

from .aligner import Backend

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SRRG2_AMD_LIB: an instrumented build of the same sources for profiling -- make OUT=... EXTRA=-DSRRG2_TILE_STATS; never a fallback)
LIB_PATH = os.environ.get("SRRG2_AMD_LIB") or os.path.join(_HERE, "lib", "libsrrg2_slam_amd.so")
_LIB = None


class LibraryMissing(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise LibraryMissing(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C srrg2_slam_interfaces_amd/csrc` (hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
        _LIB = C.CDLL(LIB_PATH)
        _LIB.srrg2_amd_last_error.restype = C.c_char_p
        _LIB.srrg2_aligner_profile_get.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]
    return _LIB


def backend():
    l = lib()
    return Backend(l, "srrg2_aligner_", l.srrg2_amd_last_error, needs_device=True)


def device_count():
    return lib().srrg2_amd_device_count()
