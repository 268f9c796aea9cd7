"""Re-statement of the reference's own tests for this path:

  T/test_motion_model.cpp:14-315        9 constant-velocity motion-model tests
  T/test_motion_model_slice.cpp:44-227  MultiAlignerSliceMotionModel3D.{Random, LocalMapCreation, Relocalization}

The slice tests drive MultiAligner3DQR::compute() with a single AlignerSliceMotionModel3D (prior-only slice,
min_num_inliers = 0) through the tracker sequence of MockedMultiTracker3D; the tracker plumbing is restated in
_MockedTracker below (S/trackers/multi_tracker_impl.cpp:57-140, tracker_slice_processor_estimation_buffer.hpp:25-79).
The CPU leg runs against the oracle, the `gpu` leg runs the same scenarios through the HIP library's C ABI.
Eigen's Vector3f::Random()/rand() are replaced by the repo's SplitMix64 stream (same distribution class:
translation in [-1,1]^3, arbitrary rotation angles)."""
import math

import numpy as np
import pytest

from srrg2_slam_interfaces_amd import _abi as abi
from srrg2_slam_interfaces_amd import slices as sl
from srrg2_slam_interfaces_amd import synthetic as syn


def _random_motion(i, dim=3):
    u = syn.uniform(9000 + i, 6)
    if dim == 3:
        return syn.se3(2 * u[:3] - 1, (2 * u[3:] - 1) * math.pi).astype(np.float32)
    return syn.se2(2 * u[0] - 1, 2 * u[1] - 1, (2 * u[2] - 1) * math.pi).astype(np.float32)


def _t2v_norm(T):
    """|t2v(T)|: translation norm and rotation magnitude (quaternion imaginary part), as geometry3d::t2v."""
    T = np.asarray(T, np.float64)
    if T.shape == (3, 3):
        return math.sqrt(T[0, 2] ** 2 + T[1, 2] ** 2 + math.atan2(T[1, 0], T[0, 0]) ** 2)
    R = T[:, :3]
    w = math.sqrt(max(1.0 + np.trace(R), 1e-12)) / 2  # near identity in every use below
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (4 * w)
    return math.sqrt(float(T[:, 3] @ T[:, 3]) + float(v @ v))


# ---- T/test_motion_model.cpp -----------------------------------------------------------------------
def _step(i, prev, kind):
    if kind == "still":
        return sl.identity(3)
    cur = prev.copy()
    if kind in ("translation", "both"):
        cur[:, 3] += 1.0
    if kind in ("rotation", "both"):
        cur = sl.compose(cur, syn.se3(np.zeros(3), np.array([0.1 * math.pi, 0, 0])).astype(np.float32))
    if kind == "random":
        cur = sl.compose(prev, _random_motion(i))
    return cur


@pytest.mark.parametrize("kind", ["still", "translation", "rotation", "both", "random"])
def test_motion_model_constant_velocity_3d(kind):
    """Still / UniformTranslation / UniformRotation / UniformTranslationAndRotation / Random (:14-121)."""
    model = sl.MotionModelConstantVelocity(3)
    prev = sl.identity(3)
    for i in range(10):
        cur = _step(i, prev, kind)
        model.set_robot_in_local_map(cur)
        model.compute()
        true_motion = sl.compose(sl.inverse(prev), cur)
        assert np.array_equal(true_motion, model.estimate())  # identical float32 expression -> exact
        prev = cur


@pytest.mark.parametrize("dim", [3, 2])
@pytest.mark.parametrize("relocalize", [False, True])
def test_motion_model_new_local_map_and_relocalization(dim, relocalize):
    """NewLocalMap / RelocalizationInLocalMap for 3D (:123-223) and 2D (:225-315)."""
    model = sl.MotionModelConstantVelocity(dim)
    prev = sl.identity(dim)
    for i in range(5):
        cur = sl.compose(prev, _random_motion(i, dim))
        model.set_robot_in_local_map(cur)
        model.compute()
        assert np.allclose(sl.compose(sl.inverse(prev), cur), model.estimate(), atol=1e-6)
        prev = cur
    prev = _random_motion(77, dim) if relocalize else sl.identity(dim)
    model.shift_tracker_estimate(prev)
    for i in range(5, 10):
        cur = sl.compose(prev, _random_motion(i, dim))
        model.set_robot_in_local_map(cur)
        model.compute()
        assert np.allclose(sl.compose(sl.inverse(prev), cur), model.estimate(), atol=1e-5)
        prev = cur


# ---- T/test_motion_model_slice.cpp -----------------------------------------------------------------
class _MockedTracker:
    """MockedMultiTracker3D + TrackerSliceProcessorEstimationBuffer3D + RawDataPreprocessorTrackerEstimate3D."""

    def __init__(self, aligner):
        self.aligner = aligner
        self.aligner.set_params(max_iterations=10, min_num_inliers=0)  # test_motion_model_slice.cpp:280
        self.slice = sl.AlignerSliceMotionModel(aligner, sl.MotionModelConstantVelocity(3))
        self.adaptor = sl.TrackerEstimateBuffer(3)
        self.robot_in_local_map = sl.identity(3)
        self._scene_changed = False
        self._relocalized = False
        self._updated = sl.identity(3)
        self.measurement = []
        self.status = "Initializing"

    def set_scene(self):  # estimation_buffer.hpp:25-35
        if not self._relocalized:
            self._updated = self.robot_in_local_map
            self._scene_changed = True
        self.status = "Tracking"

    def set_closure(self, robot_in_moving_local_map):  # :37-50
        self._updated = sl.compose(self.robot_in_local_map, sl.inverse(robot_in_moving_local_map))
        self._relocalized = True

    def set_robot_in_local_map(self, T):
        self.robot_in_local_map = np.asarray(T, np.float32)

    def preprocess_raw_data(self):  # multi_tracker_impl.cpp:57-80 -> adaptor->compute()
        self.measurement = self.adaptor.compute()
        self.status = "Tracking"

    def align(self):  # MockedMultiTracker3D::align, test_motion_model_slice.cpp:232-255
        self.slice.set_fixed(self.measurement)
        self.slice.init()
        st = self.aligner.compute()
        self.status = "Tracking" if st == abi.SUCCESS else "Lost"

    def merge(self):  # estimation_buffer.hpp:59-79
        if self._scene_changed or self._relocalized:
            self.adaptor.set_coordinate_frame_origin(self._updated)
            self._scene_changed = False
            self._relocalized = False
        self.adaptor.set_robot_in_local_map(self.robot_in_local_map)

    def compute(self):  # TrackerBase::compute, tracker.cpp:9-13
        self.preprocess_raw_data()
        self.align()
        self.merge()


def _make_aligner(backend_name, oracle):
    if backend_name == "oracle":
        return oracle.OracleAligner(abi.SE3_QUAT_RIGHT)  # MultiAligner3DQR
    import srrg2_slam_interfaces_amd as pkg

    return pkg.MultiAligner(abi.SE3_QUAT_RIGHT)


BACKENDS = ["oracle", pytest.param("hip", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("backend", BACKENDS)
def test_slice_motion_model_random(backend, oracle):
    tr = _MockedTracker(_make_aligner(backend, oracle))
    prev_pose = sl.identity(3)
    tr.set_scene()
    tr.set_robot_in_local_map(prev_pose)
    tr.compute()
    motion_previous = sl.identity(3)
    for i in range(10):
        motion = _random_motion(i)
        pose = sl.compose(prev_pose, motion)
        tr.set_robot_in_local_map(pose)
        tr.compute()
        assert tr.status == "Tracking"
        err = _t2v_norm(sl.compose(tr.aligner.moving_in_fixed(), motion_previous))
        assert err < 1e-5
        prev_pose, motion_previous = pose, motion


@pytest.mark.parametrize("backend", BACKENDS)
def test_slice_motion_model_local_map_creation(backend, oracle):
    tr = _MockedTracker(_make_aligner(backend, oracle))
    previous_estimate = sl.identity(3)
    tr.set_scene()
    tr.set_robot_in_local_map(previous_estimate)
    tr.compute()
    motion_previous = sl.identity(3)
    for i in range(100):
        motion = _random_motion(i)
        current = sl.compose(previous_estimate, motion)
        tr.preprocess_raw_data()
        tr.align()
        if i % 10 == 0:  # simulate consecutive new local map creations
            current = motion
            tr.set_scene()
        tr.set_robot_in_local_map(current)
        tr.merge()
        assert tr.status == "Tracking"
        assert _t2v_norm(sl.compose(tr.aligner.moving_in_fixed(), motion_previous)) < 1e-5
        previous_estimate, motion_previous = current, motion


@pytest.mark.parametrize("backend", BACKENDS)
def test_slice_motion_model_relocalization(backend, oracle):
    tr = _MockedTracker(_make_aligner(backend, oracle))
    origin_a = sl.identity(3)
    tr.set_scene()
    tr.set_robot_in_local_map(origin_a)
    tr.compute()
    previous_estimate = sl.identity(3)
    motion_previous = sl.identity(3)
    for i in range(100):
        motion = _random_motion(i)
        current = sl.compose(previous_estimate, motion)
        tr.preprocess_raw_data()
        tr.set_robot_in_local_map(current)
        tr.align()
        if i % 10 == 0:  # relocalize into another local map b
            origin_b = _random_motion(1000 + i)
            a_in_b = sl.compose(sl.inverse(origin_b), origin_a)
            current = sl.compose(a_in_b, current)
            tr.set_closure(current)
            tr.set_scene()
        tr.set_robot_in_local_map(current)
        tr.merge()
        assert tr.status == "Tracking"
        assert _t2v_norm(sl.compose(tr.aligner.moving_in_fixed(), motion_previous)) < 1e-4
        previous_estimate, motion_previous = current, motion


@pytest.mark.parametrize("backend", BACKENDS)
def test_odometry_prior_slice_count_semantics(backend, oracle):
    """AlignerSliceOdom3DPrior: identity measurement for the first two computes, then fixed^-1 * moving
    (aligner_slice_odometry_prior.cpp:23-37)."""
    al = _make_aligner(backend, oracle)
    al.set_params(min_num_inliers=0)
    pr = sl.AlignerSliceOdomPrior(al)
    a, b = _random_motion(1), _random_motion(2)
    pr.set_fixed(a)
    pr.set_moving(b)
    expected = [sl.identity(3), sl.identity(3), sl.compose(sl.inverse(a), b)]
    for k in range(3):
        pr.init()
        assert al.compute() == abi.SUCCESS
        assert np.allclose(al.moving_in_fixed(), expected[k], atol=2e-6)
