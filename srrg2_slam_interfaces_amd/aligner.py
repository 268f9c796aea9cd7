"""Python host-side mirror of the reference aligner interface over the C ABI.

``MultiAligner`` follows ``MultiAlignerBase_`` (S/registration/aligners/multi_aligner.h:19-150,
multi_aligner_impl.cpp:8-303): same method names (snake_case), argument meaning and error
behaviour (misuse raises ``RuntimeError`` where the reference throws ``std::runtime_error``;
algorithmic outcomes are ``status`` values 0..3 of aligner.h:23-28).

The class is a thin marshalling layer: every call goes straight through the C ABI declared in
include/srrg2_slam_amd.h.  It is parametrised by a *backend* (a loaded shared library plus the
symbol prefix) so that the test-side oracle binding can reuse the marshalling code; the product
backend is ``_capi.backend()`` and raises if the HIP library is missing.
"""
import ctypes as C

import numpy as np

from . import _abi as abi


class Backend:
    """A loaded C library exporting the aligner call surface under ``prefix``."""

    def __init__(self, lib, prefix, err_fn, needs_device):
        self.lib = lib
        self.prefix = prefix
        self._err = err_fn
        self.needs_device = needs_device

    def fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def last_error(self):
        msg = self._err()
        return msg.decode() if msg else ""


def _as_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


_BATCH_DTYPE = np.dtype(abi.BatchResult)


class BatchResults:
    """The K result records of one compute_batch*() call: a read-only sequence of dicts (``res[k]["status"]`` ...)
    over the array of ``srrg2_batch_result`` the library filled, plus whole-batch views of it (``res.status``,
    ``res.moving_in_fixed`` ...).  Nothing is copied or converted until it is asked for: a loop-closure detector
    that only reads the statuses of 256 candidates does not pay for 256 dicts."""

    def __init__(self, raw, K, dim, tsize):
        self._raw = raw  # (keeps the ctypes array alive)
        self._K = K
        self._dim = dim
        self._tsize = tsize
        self._arr = np.frombuffer(raw, dtype=_BATCH_DTYPE, count=max(K, 1))[:K]

    def __len__(self):
        return self._K

    @property
    def status(self):
        return self._arr["status"]

    @property
    def num_iterations(self):
        return self._arr["num_iterations"]

    @property
    def num_correspondences(self):
        return self._arr["num_correspondences"]

    @property
    def moving_in_fixed(self):
        T = self._arr["moving_in_fixed"][:, :self._tsize]
        return T.reshape(self._K, 3, 3) if self._dim == 2 else T.reshape(self._K, 3, 4)

    @property
    def information(self):
        D = 3 if self._dim == 2 else 6
        return self._arr["information"][:, :D * D].reshape(self._K, D, D)

    @property
    def last(self):
        """structured array of the last IterationStats of every alignment"""
        return self._arr["last"]

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(self._K))]
        if k < 0:
            k += self._K
        if not 0 <= k < self._K:
            raise IndexError(k)
        row = self._arr[k]
        D = 3 if self._dim == 2 else 6
        T = np.array(row["moving_in_fixed"][:self._tsize], dtype=np.float32)
        last = row["last"]
        return {
            "moving_in_fixed": T.reshape(3, 3) if self._dim == 2 else T.reshape(3, 4),
            "status": int(row["status"]),
            "num_iterations": int(row["num_iterations"]),
            "last": {name: last[name].item() for name in last.dtype.names},
            # numCorrespondences() after compute() (after pruning) and H of the last Gauss-Newton iteration
            "num_correspondences": int(row["num_correspondences"]),
            "information": np.array(row["information"][:D * D], dtype=np.float32).reshape(D, D),
        }

    def __iter__(self):
        return (self[k] for k in range(self._K))


class MultiAligner:
    """One ``MultiAlignerBase_<Variable>``: one estimate, N slices, one Gauss-Newton solver."""

    def __init__(self, backend, variable_kind=abi.SE3_QUAT_RIGHT, device=0):
        self._b = backend
        self.variable_kind = variable_kind
        self.dim = abi.point_dim(variable_kind)
        self.tsize = abi.transform_size(variable_kind)
        self._h = C.c_void_p()
        if backend.needs_device:
            rc = backend.fn("create")(C.c_int(variable_kind), C.c_int(device), C.byref(self._h))
        else:
            rc = backend.fn("create")(C.c_int(variable_kind), C.byref(self._h))
        self._check(rc)
        self.slices = []

    # -- plumbing -----------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("%s (code %d)" % (self._b.last_error(), rc))

    def close(self):
        if self._h:
            self._b.fn("destroy")(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration (PARAMs) -----------------------------------------------------
    def set_params(self, max_iterations=10, min_num_inliers=10, enable_inlier_only_runs=False,
                   keep_only_inlier_correspondences=False):
        p = abi.AlignerParams(max_iterations, min_num_inliers, int(enable_inlier_only_runs),
                              int(keep_only_inlier_correspondences))
        self._check(self._b.fn("set_params")(self._h, C.byref(p)))

    def set_termination_criteria(self, params):
        """``params``: abi.TerminationParams or None (param_termination_criteria, aligner.h:31-35)."""
        if params is None:
            self._check(self._b.fn("set_termination")(self._h, None))
        else:
            self._check(self._b.fn("set_termination")(self._h, C.byref(params)))

    def tuning(self):
        """the handle's strategy knobs (abi.AlignerTuning): defaults overridden by the SRRG2_AMD_* environment at creation"""
        t = abi.AlignerTuning()
        self._check(self._b.fn("get_tuning")(self._h, C.byref(t)))
        return t

    def share_clouds(self, slice_idx, source_slice_idx):
        """slice ``slice_idx`` reads the fixed and moving clouds of ``source_slice_idx`` (no copy; -1: its own again): the
        reference's slices find their clouds by name in the scene, two slices with the same names bind to the same clouds
        (aligner_slice_processor_base_impl.cpp:27-50)"""
        self._check(self._b.fn("share_clouds")(self._h, C.c_int(slice_idx), C.c_int(source_slice_idx)))

    def set_tuning(self, **knobs):
        """change strategy knobs by name (fields of srrg2_aligner_tuning); results do not depend on them"""
        t = self.tuning()
        for k, v in knobs.items():
            if k not in dict(abi.AlignerTuning._fields_) or k == "reserved_":
                raise KeyError(k)
            setattr(t, k, v)
        self._check(self._b.fn("set_tuning")(self._h, C.byref(t)))

    def last_compute_path(self):
        """SRRG2_PATH_* bits of the launch path the last compute() took (strategy only; 0 for a backend that has no such notion)"""
        try:
            f = self._b.fn("last_compute_path")
        except AttributeError:
            return 0
        v = C.c_int32(0)
        self._check(f(self._h, C.byref(v)))
        return int(v.value)

    def add_slice(self, config):
        idx = C.c_int(-1)
        self._check(self._b.fn("add_slice")(self._h, C.byref(config), C.byref(idx)))
        self.slices.append(config)
        return idx.value

    def clear_slices(self):
        self._check(self._b.fn("clear_slices")(self._h))
        self.slices = []

    def set_robustifier(self, slice_idx, kind, chi_threshold):
        self._check(self._b.fn("set_robustifier")(self._h, C.c_int(slice_idx), C.c_int(kind),
                                                  C.c_float(chi_threshold)))

    # -- data ---------------------------------------------------------------------
    def _set_cloud(self, which, slice_idx, coords, normals):
        coords = _as_f32(coords)
        assert coords.ndim == 2 and coords.shape[1] == self.dim, coords.shape
        n = coords.shape[0]
        nptr, nstride = None, 0
        if normals is not None:
            normals = _as_f32(normals)
            assert normals.shape == coords.shape
            nptr, nstride = _fptr(normals), normals.strides[0]
        self._check(self._b.fn(which)(self._h, C.c_int(slice_idx), _fptr(coords), C.c_int(coords.strides[0]),
                                      nptr, C.c_int(nstride), C.c_int(n), C.c_int(abi.MEM_HOST)))

    def set_fixed(self, slice_idx, coords, normals=None):
        self._set_cloud("set_fixed", slice_idx, coords, normals)

    def set_moving(self, slice_idx, coords, normals=None):
        self._set_cloud("set_moving", slice_idx, coords, normals)

    def set_cloud_device(self, which, slice_idx, coords_ptr, coord_stride, normals_ptr, normal_stride, n, kept=False):
        """Device-resident input (``which`` = 'set_fixed' | 'set_moving'); pointers are raw ints.  ``kept``: the buffer stays
        valid and unchanged until the next compute() has returned (SRRG2_MEM_DEVICE_KEPT: the call does not wait for the ingest)."""
        self._check(self._b.fn(which)(self._h, C.c_int(slice_idx), C.cast(coords_ptr, C.POINTER(C.c_float)),
                                      C.c_int(coord_stride),
                                      C.cast(normals_ptr, C.POINTER(C.c_float)) if normals_ptr else None,
                                      C.c_int(normal_stride), C.c_int(n), C.c_int(abi.MEM_DEVICE_KEPT if kept else abi.MEM_DEVICE)))

    def set_sensor_in_robot(self, slice_idx, T):
        """slice->setSensorInRobot with the transform looked up on every setMovingInFixed
        (aligner_slice_processor_impl.cpp:20-36); takes effect from the next compute() on."""
        T = _as_f32(T).reshape(-1)
        assert T.size == self.tsize
        self._check(self._b.fn("set_sensor_in_robot")(self._h, C.c_int(slice_idx), _fptr(T)))

    REDUCE_SUM_I64, REDUCE_MAX_U32 = 0, 1
    _REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p)

    def set_point_shard(self, reduce, total_moving_points):
        """ONE alignment sharded by moving points over the ranks of a process group (srrg2_aligner_set_point_shard):
        this rank sets its share of the moving cloud, every compute() calls ``reduce(op, device_ptr, count, stream)``
        -- in place, on all ranks alike -- for the max |coordinate| word (op REDUCE_MAX_U32, once) and for the int64
        partial sums before every control step (op REDUCE_SUM_I64).  The estimate, the statistics and H are then those of
        the one-GPU alignment of the whole cloud, bit for bit.  ``reduce=None`` switches the mode off.
        (distributed.point_shard_reducer builds ``reduce`` on torch.distributed)"""
        if reduce is None:
            self._shard_cb = None
            self._check(self._b.fn("set_point_shard")(self._h, None, None, C.c_int64(0)))
            return

        def _cb(_user, op, ptr, count, stream):
            try:
                reduce(int(op), int(ptr), int(count), int(stream) if stream else 0)
                return 0
            except Exception:  # (an exception must not unwind through the C frames)
                import traceback

                traceback.print_exc()
                return 1

        self._shard_cb = self._REDUCE_FN(_cb)  # (kept alive with the handle)
        self._check(self._b.fn("set_point_shard")(self._h, self._shard_cb, None, C.c_int64(int(total_moving_points))))

    def set_prior_measurement(self, slice_idx, T):
        T = _as_f32(T).reshape(-1)
        assert T.size == self.tsize
        self._check(self._b.fn("set_prior_measurement")(self._h, C.c_int(slice_idx), _fptr(T)))

    def set_moving_in_fixed(self, T):
        T = _as_f32(T).reshape(-1)
        assert T.size == self.tsize
        self._check(self._b.fn("set_moving_in_fixed")(self._h, _fptr(T)))

    def moving_in_fixed(self):
        T = np.zeros(self.tsize, dtype=np.float32)
        self._check(self._b.fn("get_moving_in_fixed")(self._h, _fptr(T)))
        return T.reshape(3, 3) if self.dim == 2 else T.reshape(3, 4)

    # -- compute --------------------------------------------------------------------
    def compute(self):
        st = C.c_int(-1)
        self._check(self._b.fn("compute")(self._h, C.byref(st)))
        return st.value

    def status(self):
        st = C.c_int(-1)
        self._check(self._b.fn("status")(self._h, C.byref(st)))
        return st.value

    def iteration_stats(self):
        n = C.c_int(0)
        self._check(self._b.fn("get_iteration_stats")(self._h, None, C.byref(n)))
        buf = (abi.IterationStats * max(n.value, 1))()
        n2 = C.c_int(n.value)
        self._check(self._b.fn("get_iteration_stats")(self._h, buf, C.byref(n2)))
        return [buf[i].as_dict() for i in range(n2.value)]

    def information(self):
        """H = sum w J^T J of the last Gauss-Newton iteration of the last compute(), D x D float32 (product backend)"""
        D = 3 if self.variable_kind == abi.SE2_RIGHT else 6
        H = np.zeros((D, D), np.float32)
        self._check(self._b.fn("get_information")(self._h, _fptr(H)))
        return H

    def last_iteration_stats(self):
        """(number of IterationStats of the last compute(), the last one as a dict): what the batch callers' gates read
        (multi_loop_detector_brute_force_impl.cpp:80-91), without building one dict per iteration."""
        buf = getattr(self, "_stats_buf", None)
        if buf is None:
            buf = self._stats_buf = (abi.IterationStats * 256)()
        n = C.c_int(256)
        self._check(self._b.fn("get_iteration_stats")(self._h, buf, C.byref(n)))
        if n.value > 256:  # (more than the scratch holds: take the general path)
            stats = self.iteration_stats()
            return len(stats), stats[-1]
        return n.value, (buf[n.value - 1].as_dict() if n.value > 0 else None)

    def num_correspondences(self):
        n = C.c_int(0)
        self._check(self._b.fn("num_correspondences")(self._h, C.byref(n)))
        return n.value

    def correspondences(self, slice_idx):
        """(C,) structured array with fields fixed_idx, moving_idx, response."""
        n = C.c_int(0)
        self._check(self._b.fn("get_correspondences")(self._h, C.c_int(slice_idx), None, C.byref(n)))
        dt = np.dtype([("fixed_idx", np.int32), ("moving_idx", np.int32), ("response", np.float32)])
        out = np.zeros(max(n.value, 1), dtype=dt)
        n2 = C.c_int(n.value)
        self._check(self._b.fn("get_correspondences")(self._h, C.c_int(slice_idx),
                                                      out.ctypes.data_as(C.POINTER(abi.Correspondence)),
                                                      C.byref(n2)))
        return out[:n2.value]

    def factor_status(self, slice_idx):
        n = C.c_int(0)
        self._check(self._b.fn("get_factor_status")(self._h, C.c_int(slice_idx), None, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.uint8)
        n2 = C.c_int(n.value)
        self._check(self._b.fn("get_factor_status")(self._h, C.c_int(slice_idx),
                                                    out.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(n2)))
        return out[:n2.value]

    _CORR_DTYPE = np.dtype([("fixed_idx", np.int32), ("moving_idx", np.int32), ("response", np.float32)])

    @classmethod
    def _corr_array(cls, corr):
        """(pointer, keep-alive array): srrg2_correspondence records from a structured array or a list of dicts"""
        if isinstance(corr, np.ndarray) and corr.dtype == cls._CORR_DTYPE and corr.flags.c_contiguous and corr.size:
            a = corr  # (already srrg2_correspondence records: borrowed for the call, not copied)
        elif isinstance(corr, np.ndarray) and corr.dtype.names:
            a = np.empty(len(corr), cls._CORR_DTYPE)
            for f in cls._CORR_DTYPE.names:
                a[f] = corr[f]
        else:
            a = np.array([(int(c["fixed_idx"]), int(c["moving_idx"]), float(c["response"])) for c in corr], cls._CORR_DTYPE)
        if a.size == 0:
            a = np.zeros(1, cls._CORR_DTYPE)
        return a.ctypes.data_as(C.POINTER(abi.Correspondence)), a

    def set_correspondences(self, slice_idx, corr):
        """factor->setCorrespondences(corrs) for a FINDER_CORRESPONDENCES slice: the pairs stay locked during
        compute() (multi_loop_detector_hbst_impl.cpp:330,343).  corr: structured array / list of dicts."""
        ptr, keep = self._corr_array(corr)
        self._check(self._b.fn("set_correspondences")(self._h, C.c_int(slice_idx), ptr, C.c_int(len(corr))))

    def compute_batch_correspondences(self, moving_clouds, correspondences, guesses, moving_normals=None):
        """K alignments with given correspondences against the fixed cloud (the loop of
        MultiLoopDetectorHBST_::_computeAlignments, multi_loop_detector_hbst_impl.cpp:296-374)."""
        K = len(moving_clouds)
        offsets = np.zeros(K + 1, dtype=np.int32)
        offsets[1:] = np.cumsum([int(np.asarray(m).shape[0]) for m in moving_clouds])
        coffsets = np.zeros(K + 1, dtype=np.int32)
        coffsets[1:] = np.cumsum([len(c) for c in correspondences])
        coords = _as_f32(np.concatenate([_as_f32(m) for m in moving_clouds], axis=0)) if K else np.zeros((0, self.dim), np.float32)
        nptr, nstride = None, 0
        if moving_normals is not None:
            normals = _as_f32(np.concatenate([_as_f32(m) for m in moving_normals], axis=0))
            nptr, nstride = _fptr(normals), normals.strides[0]
        parts = [self._corr_array(cs)[1][: len(cs)] for cs in correspondences]
        # (concatenated as plain int32 words: numpy copies structured records field by field, 6x slower)
        cptr, keep = self._corr_array(np.concatenate([q.view(np.int32) for q in parts]).view(self._CORR_DTYPE)
                                      if parts else np.zeros(0, self._CORR_DTYPE))
        g = _as_f32(np.asarray(guesses)).reshape(K, self.tsize)
        res = (abi.BatchResult * max(K, 1))()
        self._check(self._b.fn("compute_batch_correspondences")(
            self._h, C.c_int(K), _fptr(coords), C.c_int(coords.strides[0]), nptr, C.c_int(nstride),
            offsets.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int(abi.MEM_HOST), cptr,
            coffsets.ctypes.data_as(C.POINTER(C.c_int32)), _fptr(g), res))
        return self._unpack_batch(res, K)

    def compute_batch_device(self, coords_ptr, coord_stride, normals_ptr, normal_stride, offsets, guesses):
        """compute_batch on clouds already resident in HBM (raw device pointers as ints)."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        K = offsets.size - 1
        g = _as_f32(np.asarray(guesses)).reshape(K, self.tsize)
        res = (abi.BatchResult * max(K, 1))()
        self._check(self._b.fn("compute_batch")(
            self._h, C.c_int(K), C.cast(coords_ptr, C.POINTER(C.c_float)), C.c_int(coord_stride),
            C.cast(normals_ptr, C.POINTER(C.c_float)) if normals_ptr else None, C.c_int(normal_stride),
            offsets.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int(abi.MEM_DEVICE), _fptr(g), res))
        return self._unpack_batch(res, K)

    def _unpack_batch(self, res, K):
        return BatchResults(res, K, self.dim, self.tsize)

    def compute_batch(self, moving_clouds, guesses, moving_normals=None):
        """K independent alignments against the fixed scene (multi_loop_detector_brute_force_impl.cpp:63-91)."""
        K = len(moving_clouds)
        sizes = [int(np.asarray(m).shape[0]) for m in moving_clouds]
        offsets = np.zeros(K + 1, dtype=np.int32)
        offsets[1:] = np.cumsum(sizes)
        coords = _as_f32(np.concatenate([_as_f32(m) for m in moving_clouds], axis=0)) if K else np.zeros(
            (0, self.dim), np.float32)
        nptr, nstride = None, 0
        if moving_normals is not None:
            normals = _as_f32(np.concatenate([_as_f32(m) for m in moving_normals], axis=0))
            nptr, nstride = _fptr(normals), normals.strides[0]
        g = _as_f32(np.asarray(guesses)).reshape(K, self.tsize)
        res = (abi.BatchResult * max(K, 1))()
        self._check(self._b.fn("compute_batch")(self._h, C.c_int(K), _fptr(coords), C.c_int(coords.strides[0]),
                                                nptr, C.c_int(nstride),
                                                offsets.ctypes.data_as(C.POINTER(C.c_int32)),
                                                C.c_int(abi.MEM_HOST), _fptr(g), res))
        return self._unpack_batch(res, K)
