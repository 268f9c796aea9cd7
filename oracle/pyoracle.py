"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from srrg2_slam_interfaces_amd import _abi as abi
from srrg2_slam_interfaces_amd.aligner import Backend, MultiAligner

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _LIB
    if _LIB is None:
        # SRRG2_ORACLE_LIB: another build of the same sources (bench.py's cpu_baseline leg compiles one with
        # -march=native on the box it times)
        path = os.environ.get("SRRG2_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.oracle_last_error.restype = C.c_char_p
        _LIB.o_atan2.restype = C.c_double
        _LIB.o_atan2.argtypes = [C.c_double, C.c_double]
        _LIB.o_sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _LIB.o_fixed_point_exponent.argtypes = [C.c_int, C.c_double]
    return _LIB


def backend():
    l = lib()
    return Backend(l, "oracle_aligner_", l.oracle_last_error, needs_device=False)


class OracleAligner(MultiAligner):
    def __init__(self, variable_kind=abi.SE3_QUAT_RIGHT):
        super().__init__(backend(), variable_kind)

    def set_bruteforce(self, enable):
        self._check(lib().oracle_aligner_set_bruteforce(self._h, C.c_int(int(enable))))

    def linearize_once(self, slice_idx):
        acc = np.zeros(32, dtype=np.int64)
        k = C.c_int(0)
        self._check(lib().oracle_aligner_linearize_once(self._h, C.c_int(slice_idx),
                                                        acc.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(k)))
        return acc, k.value

    def last_system(self):
        D = 3 if self.dim == 2 else 6
        H = np.zeros((D, D))
        b = np.zeros(D)
        dx = np.zeros(D)
        dp = C.POINTER(C.c_double)
        self._check(lib().oracle_aligner_get_last_system(self._h, H.ctypes.data_as(dp), b.ctypes.data_as(dp),
                                                         dx.ctypes.data_as(dp)))
        return H, b, dx


# ---- thin wrappers of the oracle's math, used by the unit tests -------------------
def sincos(x):
    s, c = C.c_double(), C.c_double()
    lib().o_sincos(C.c_double(x), C.byref(s), C.byref(c))
    return s.value, c.value


def atan2(y, x):
    return lib().o_atan2(y, x)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def se3_compose(A, B):
    A, B = _f(A).reshape(-1), _f(B).reshape(-1)
    out = np.zeros(12, np.float32)
    lib().o_se3_compose(_fp(A), _fp(B), _fp(out))
    return out.reshape(3, 4)


def se3_inverse(A):
    A = _f(A).reshape(-1)
    out = np.zeros(12, np.float32)
    lib().o_se3_inverse(_fp(A), _fp(out))
    return out.reshape(3, 4)


def se2_compose(A, B):
    A, B = _f(A).reshape(-1), _f(B).reshape(-1)
    out = np.zeros(9, np.float32)
    lib().o_se2_compose(_fp(A), _fp(B), _fp(out))
    return out.reshape(3, 3)


def se2_inverse(A):
    A = _f(A).reshape(-1)
    out = np.zeros(9, np.float32)
    lib().o_se2_inverse(_fp(A), _fp(out))
    return out.reshape(3, 3)


def se3_v2t(kind, v):
    v = np.ascontiguousarray(v, dtype=np.float64)
    R = np.zeros(9)
    t = np.zeros(3)
    lib().o_se3_v2t(C.c_int(kind), _dp(v), _dp(R), _dp(t))
    return R.reshape(3, 3), t


def se3_t2v_quat(T):
    T = _f(T).reshape(-1)
    v = np.zeros(6)
    lib().o_se3_t2v_quat(_fp(T), _dp(v))
    return v


def se2_t2v(T):
    T = _f(T).reshape(-1)
    v = np.zeros(3)
    lib().o_se2_t2v(_fp(T), _dp(v))
    return v


def box_plus(kind, X, dx):
    X = _f(X).reshape(-1).copy()
    dx = np.ascontiguousarray(dx, dtype=np.float64)
    lib().o_box_plus(C.c_int(kind), _fp(X), _dp(dx))
    return X.reshape(3, 3) if kind == abi.SE2_RIGHT else X.reshape(3, 4)


def fix_transform(kind, T):
    T = _f(T).reshape(-1).copy()
    if kind == abi.SE2_RIGHT:
        lib().o_se2_fix_transform(_fp(T))
        return T.reshape(3, 3)
    lib().o_se3_fix_transform(_fp(T))
    return T.reshape(3, 4)


def solve(H, b):
    H = np.ascontiguousarray(H, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    D = b.size
    dx = np.zeros(D)
    rc = lib().o_solve(C.c_int(D), _dp(H), _dp(b), _dp(dx))
    return rc, dx


# ---- pose graph -------------------------------------------------------------------------------
from srrg2_slam_interfaces_amd.posegraph import PoseGraph as _PoseGraph  # noqa: E402


class OraclePoseGraph(_PoseGraph):
    def __init__(self, variable_kind=abi.SE3_QUAT_RIGHT):
        l = lib()
        l.oracle_posegraph_chi.restype = C.c_double
        super().__init__(l, "oracle_posegraph_", l.oracle_last_error, variable_kind, device=None)

    def set_direct(self, enable):
        self._check(lib().oracle_posegraph_set_direct(self._h, C.c_int(int(enable))))

    def chi(self):
        return lib().oracle_posegraph_chi(self._h)

    def edge(self, e):
        D = self.D
        err, Ji, Jj = np.zeros(D), np.zeros((D, D)), np.zeros((D, D))
        self._check(lib().oracle_posegraph_edge(self._h, C.c_int(e), _dp(err), _dp(Ji), _dp(Jj)))
        return err, Ji, Jj


# ---- scene clipping / merging --------------------------------------------------------------------
from srrg2_slam_interfaces_amd.mapping import _Binding as _SceneBinding  # noqa: E402


def scene_binding():
    l = lib()
    return _SceneBinding(l, "oracle_scene_", l.oracle_last_error, None)
