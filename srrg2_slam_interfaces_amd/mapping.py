"""Python mirror of the tracker-side scene steps over the C ABI (srrg2_scene_*).

``Scene``                      a point(+normal) cloud kept in device memory (a LocalMap scene slice / a measurement)
``SceneClipperBall``           ``SceneClipper_`` (S/mapping/scene_clipper.h:17-122) with the ball policy
``MergerCorrespondenceHomo``   ``MergerCorrespondenceHomo_`` (S/mapping/merger_correspondence_homo_impl.cpp:11-125)

Method names follow the reference setters (snake_case).  Thin marshalling only; parametrised by
(lib, prefix, err_fn) like ``posegraph.PoseGraph`` so that the test-side oracle binding reuses it.
"""
import ctypes as C

import numpy as np

from . import _abi as abi

MERGER_ERROR, MERGER_INITIALIZING, MERGER_SUCCESS = 0, 1, 2
CLIPPER_ERROR, CLIPPER_SUCCESSFUL, CLIPPER_READY = 0, 1, 2


class MergerParams(C.Structure):
    _fields_ = [("maximum_response", C.c_float), ("maximum_distance_geometry_squared", C.c_float),
                ("target_number_of_merges", C.c_int32)]


class MergeResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("num_correspondences", C.c_int32), ("num_merged", C.c_int32),
                ("num_added", C.c_int32), ("scene_size", C.c_int32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def default_merger_params():
    """merger.h:126-131, merger_correspondence_homo.h:22-31"""
    return MergerParams(50.0, 0.25, 200)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class _Binding:
    def __init__(self, lib, prefix, err_fn, device):
        self.lib, self.prefix, self.err, self.device = lib, prefix, err_fn, device

    def fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def check(self, rc):
        if rc != 0:
            msg = self.err()
            raise RuntimeError("%s (code %d)" % (msg.decode() if msg else "", rc))


class Scene:
    def __init__(self, binding, dim=3):
        self._b = binding
        self.dim = dim
        self._h = C.c_void_p()
        if binding.device is None:
            rc = binding.fn("create")(C.c_int(dim), C.byref(self._h))
        else:
            rc = binding.fn("create")(C.c_int(dim), C.c_int(binding.device), C.byref(self._h))
        binding.check(rc)

    def close(self):
        if self._h:
            self._b.fn("destroy")(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set(self, coords, normals=None):
        c = _f32(coords).reshape(-1, self.dim)
        n = None if normals is None else _f32(normals).reshape(-1, self.dim)
        args = [self._h, _fp(c), C.c_int(4 * self.dim), _fp(n) if n is not None else None, C.c_int(4 * self.dim),
                C.c_int(c.shape[0])]
        if self._b.device is not None:
            args.append(C.c_int(abi.MEM_HOST))
        self._b.check(self._b.fn("set")(*args))

    def size(self):
        n = C.c_int(0)
        self._b.check(self._b.fn("size")(self._h, C.byref(n)))
        return n.value

    def get(self):
        """(coords, normals) as (n, dim) float32 arrays."""
        n = self.size()
        c = np.zeros((max(n, 1), self.dim), np.float32)
        m = np.zeros((max(n, 1), self.dim), np.float32)
        k = C.c_int(0)
        self._b.check(self._b.fn("get")(self._h, _fp(c), _fp(m), C.c_int(n), C.byref(k)))
        return c[:n], m[:n]

    def global_indices(self):
        n = C.c_int(0)
        self._b.check(self._b.fn("global_indices")(self._h, None, C.byref(n)))
        buf = np.zeros(max(n.value, 1), np.int32)
        k = C.c_int(n.value)
        self._b.check(self._b.fn("global_indices")(self._h, buf.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(k)))
        return buf[:n.value]

    def device_arrays(self):
        """(coords_ptr, normals_ptr or None, n): device float4 arrays (product backend only)."""
        c, m, n = C.POINTER(C.c_float)(), C.POINTER(C.c_float)(), C.c_int(0)
        self._b.check(self._b.fn("device_arrays")(self._h, C.byref(c), C.byref(m), C.byref(n)))
        return c, (m if m else None), n.value


class SceneClipperBall:
    """setFullScene / setClippedSceneInRobot / setRobotInLocalMap / compute / status / globalIndices."""

    def __init__(self, binding, range_max=10.0):
        self._b = binding
        self.range_max = float(range_max)
        self._full = self._clipped = None
        self._robot_in_local_map = None
        self._status = CLIPPER_ERROR

    def set_full_scene(self, scene):
        self._full = scene

    def set_clipped_scene_in_robot(self, scene):
        self._clipped = scene

    def set_robot_in_local_map(self, T):
        self._robot_in_local_map = _f32(T)

    def compute(self):
        if self._full is None or self._clipped is None or self._robot_in_local_map is None:
            raise RuntimeError("SceneClipperBall::compute|scene, output or pose not set")
        st = C.c_int(0)
        self._b.check(self._b.fn("clip_ball")(self._full._h, _fp(self._robot_in_local_map), C.c_float(self.range_max),
                                              self._clipped._h, C.byref(st)))
        self._status = st.value

    def status(self):
        return self._status

    def global_indices(self):
        return self._clipped.global_indices()


class MergerCorrespondenceHomo:
    """setScene / setMeasurement / setMeasurementInScene / setCorrespondences / compute / status."""

    def __init__(self, binding, params=None):
        self._b = binding
        self.params = params or default_merger_params()
        self._scene = self._meas = None
        self._T = None
        self._corr = None  # None = "no correspondences set" (merger_correspondence_homo_impl.cpp:30)
        self._status = MERGER_ERROR
        self.last = None

    def set_scene(self, scene):
        self._scene = scene

    def set_measurement(self, scene):
        self._meas = scene

    def set_measurement_in_scene(self, T):
        self._T = _f32(T)

    def set_correspondences(self, corr):
        """structured array with fixed_idx (scene), moving_idx (measurement), response -- or None."""
        self._corr = corr

    def _ready(self):
        if self._scene is None or self._meas is None or self._T is None:
            raise RuntimeError("MergerCorrespondenceHomo::compute|scene, measurement or transform not set")

    def compute(self):
        self._ready()
        out = MergeResult()
        if self._corr is None:
            cptr, n = None, -1
        else:
            arr = (abi.Correspondence * max(len(self._corr), 1))()
            for k, c in enumerate(self._corr):
                arr[k].fixed_idx, arr[k].moving_idx, arr[k].response = int(c["fixed_idx"]), int(c["moving_idx"]), float(c["response"])
            cptr, n = arr, len(self._corr)
        self._b.check(self._b.fn("merge")(self._scene._h, self._meas._h, _fp(self._T), cptr, C.c_int(n),
                                          C.byref(self.params), C.byref(out)))
        self._status, self.last = out.status, out.as_dict()
        return self.last

    def compute_from_aligner(self, aligner, slice_idx, clipped):
        """correspondences taken on the device from an aligner whose moving cloud was ``clipped`` and whose fixed
        cloud was the measurement (TrackerSliceProcessor_::merge(), tracker_slice_processor_impl.cpp:160-186)."""
        self._ready()
        out = MergeResult()
        self._b.check(self._b.fn("merge_from_aligner")(self._scene._h, self._meas._h, _fp(self._T), aligner._h,
                                                       C.c_int(slice_idx), clipped._h, C.byref(self.params), C.byref(out)))
        self._status, self.last = out.status, out.as_dict()
        return self.last

    def status(self):
        return self._status
