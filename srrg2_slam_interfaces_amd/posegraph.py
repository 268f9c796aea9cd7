"""Python mirror of the global pose-graph solve over the C ABI (srrg2_posegraph_*).

Replaces the call sequence of MultiGraphSLAM_::optimize() (S/system/multi_graph_slam_impl.cpp:300-317):
``graph->bindFactors(); global_solver->setGraph(graph); global_solver->compute()``.  Thin marshalling only;
parametrised by a backend like ``aligner.MultiAligner`` so that the oracle binding can reuse it.
"""
import ctypes as C

import numpy as np

from . import _abi as abi


class PoseGraphParams(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("pcg_max_iterations", C.c_int32), ("pcg_tolerance", C.c_float),
                ("damping", C.c_float)]


class PoseGraphStats(C.Structure):
    _fields_ = [("iteration", C.c_int32), ("num_factors", C.c_int32), ("pcg_iterations", C.c_int32),
                ("solver_status", C.c_int32), ("chi", C.c_float), ("pcg_residual", C.c_float)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class PoseGraphTuning(C.Structure):
    """srrg2_posegraph_tuning: strategy knobs of the solver; every setting solves to the same tolerance"""
    _fields_ = [("match_passes", C.c_int32), ("two_phase", C.c_int32), ("use_graph", C.c_int32), ("debug", C.c_int32),
                ("keep_structure", C.c_int32), ("omega_p", C.c_float), ("omega", C.c_float), ("lag_below", C.c_float),
                ("device_structure", C.c_int32), ("reserved_", C.c_int32 * 7)]


def default_params():
    return PoseGraphParams(10, 600, 1e-6, 0.0)


class PoseGraph:
    def __init__(self, lib, prefix, err_fn, variable_kind=abi.SE3_QUAT_RIGHT, device=None):
        self._lib, self._prefix, self._err = lib, prefix, err_fn
        self.variable_kind = variable_kind
        self.tsize = abi.transform_size(variable_kind)
        self.D = 3 if variable_kind == abi.SE2_RIGHT else 6
        self._h = C.c_void_p()
        if device is None:
            rc = self._fn("create")(C.c_int(variable_kind), C.byref(self._h))
        else:
            rc = self._fn("create")(C.c_int(variable_kind), C.c_int(device), C.byref(self._h))
        self._check(rc)
        self.V = self.E = 0

    def _fn(self, name):
        return getattr(self._lib, self._prefix + name)

    def _check(self, rc):
        if rc != 0:
            msg = self._err()
            raise RuntimeError("%s (code %d)" % (msg.decode() if msg else "", rc))

    def close(self):
        if self._h:
            self._fn("destroy")(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_graph(self, poses, ij, Z, omega=None, fixed_mask=None, enabled=None):
        poses = np.ascontiguousarray(poses, np.float32).reshape(-1, self.tsize)
        ij = np.ascontiguousarray(ij, np.int32).reshape(-1, 2)
        Z = np.ascontiguousarray(Z, np.float32).reshape(-1, self.tsize)
        self.V, self.E = poses.shape[0], ij.shape[0]
        assert Z.shape[0] == self.E
        fp = C.POINTER(C.c_float)
        bp = C.POINTER(C.c_uint8)
        om = None
        if omega is not None:
            omega = np.ascontiguousarray(omega, np.float32).reshape(self.E, self.D * self.D)
            om = omega.ctypes.data_as(fp)
        fm = None
        if fixed_mask is not None:
            fixed_mask = np.ascontiguousarray(fixed_mask, np.uint8)
            fm = fixed_mask.ctypes.data_as(bp)
        en = None
        if enabled is not None:
            enabled = np.ascontiguousarray(enabled, np.uint8)
            en = enabled.ctypes.data_as(bp)
        self._check(self._fn("set")(self._h, C.c_int(self.V), poses.ctypes.data_as(fp), fm, C.c_int(self.E),
                                    ij.ctypes.data_as(C.POINTER(C.c_int32)), Z.ctypes.data_as(fp), om, en))

    def tuning(self):
        """the handle's strategy knobs (PoseGraphTuning): defaults overridden by the SRRG2_AMD_PG_* environment at creation"""
        t = PoseGraphTuning()
        self._check(self._fn("get_tuning")(self._h, C.byref(t)))
        return t

    def set_tuning(self, **knobs):
        t = self.tuning()
        for k, v in knobs.items():
            if k not in dict(PoseGraphTuning._fields_) or k == "reserved_":
                raise KeyError(k)
            setattr(t, k, v)
        self._check(self._fn("set_tuning")(self._h, C.byref(t)))

    def set_enabled(self, enabled):
        enabled = np.ascontiguousarray(enabled, np.uint8)
        assert enabled.size == self.E
        self._check(self._fn("set_enabled")(self._h, enabled.ctypes.data_as(C.POINTER(C.c_uint8))))

    # -- incremental interface (product backend): the lifecycle of MultiGraphSLAM_, multi_graph_slam_impl.cpp:52-90,227-297
    def add_variable(self, pose, fixed=False):
        pose = np.ascontiguousarray(pose, np.float32).reshape(-1)
        assert pose.size == self.tsize
        vid = C.c_int(-1)
        self._check(self._fn("add_variable")(self._h, pose.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(int(fixed)),
                                             C.byref(vid)))
        self.V += 1
        return vid.value

    def add_factor(self, i, j, Z, information=None, enabled=True):
        Z = np.ascontiguousarray(Z, np.float32).reshape(-1)
        assert Z.size == self.tsize
        info = None
        if information is not None:
            information = np.ascontiguousarray(information, np.float32).reshape(self.D * self.D)
            info = information.ctypes.data_as(C.POINTER(C.c_float))
        fid = C.c_int(-1)
        self._check(self._fn("add_factor")(self._h, C.c_int(i), C.c_int(j), Z.ctypes.data_as(C.POINTER(C.c_float)), info,
                                           C.c_int(int(enabled)), C.byref(fid)))
        self.E += 1
        return fid.value

    def set_factor_enabled(self, factor_id, enabled):
        self._check(self._fn("set_factor_enabled")(self._h, C.c_int(factor_id), C.c_int(int(enabled))))

    def remove_factor(self, factor_id):
        self._check(self._fn("remove_factor")(self._h, C.c_int(factor_id)))

    def size(self):
        """(variables, factors still in the graph, enabled factors)"""
        v, f, e = C.c_int(0), C.c_int(0), C.c_int(0)
        self._check(self._fn("size")(self._h, C.byref(v), C.byref(f), C.byref(e)))
        return v.value, f.value, e.value

    def structure_info(self):
        """(structure builds of the multigrid hierarchy so far, appended leaves the last solve eliminated instead of rebuilding)"""
        b, t = C.c_int(0), C.c_int(0)
        self._check(self._fn("structure_info")(self._h, C.byref(b), C.byref(t)))
        return b.value, t.value

    def solve(self, params=None):
        params = params or default_params()
        n = C.c_int(max(params.max_iterations, 1))
        buf = (PoseGraphStats * n.value)()
        self._check(self._fn("solve")(self._h, C.byref(params), buf, C.byref(n)))
        return [buf[i].as_dict() for i in range(min(n.value, len(buf)))]

    def poses(self):
        out = np.zeros((max(self.V, 1), self.tsize), np.float32)
        self._check(self._fn("get_poses")(self._h, out.ctypes.data_as(C.POINTER(C.c_float))))
        shape = (3, 3) if self.tsize == 9 else (3, 4)
        return out[:self.V].reshape((self.V,) + shape)
