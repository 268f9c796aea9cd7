"""Host-side mirrors of the reference's prior slices and their helpers.

These classes hold the little bit of host logic that sits between a caller and
``srrg2_aligner_set_prior_measurement``; names and behaviour follow the reference:

  MotionModelConstantVelocity     S/motion_models/motion_model_constant_velocity.hpp:17-46
  TrackerEstimateBuffer           S/raw_data_preprocessors/raw_data_preprocessor_tracker_estimate.hpp:30-68
  AlignerSliceMotionModel         S/registration/aligners/aligner_slice_motion_model.hpp:44-79
  AlignerSliceOdomPrior           S/registration/aligners/aligner_slice_odometry_prior.cpp:6-37

Transforms are float32 numpy arrays: 3x3 homogeneous for SE(2), 3x4 [R|t] for SE(3) (Isometry2f/3f).
"""
import numpy as np

from . import _abi as abi


def _dim_of(T):
    return 2 if T.shape == (3, 3) else 3


def identity(dim):
    return np.eye(3, dtype=np.float32) if dim == 2 else np.eye(3, 4, dtype=np.float32)


def compose(A, B):
    A = np.asarray(A, np.float32)
    B = np.asarray(B, np.float32)
    if _dim_of(A) == 2:
        return (A @ B).astype(np.float32)
    C = np.empty((3, 4), np.float32)
    C[:, :3] = A[:, :3] @ B[:, :3]
    C[:, 3] = A[:, :3] @ B[:, 3] + A[:, 3]
    return C


def inverse(A):
    A = np.asarray(A, np.float32)
    if _dim_of(A) == 2:
        Ai = np.eye(3, dtype=np.float32)
        Ai[:2, :2] = A[:2, :2].T
        Ai[:2, 2] = -A[:2, :2].T @ A[:2, 2]
        return Ai
    Ai = np.empty((3, 4), np.float32)
    Ai[:, :3] = A[:, :3].T
    Ai[:, 3] = -A[:, :3].T @ A[:, 3]
    return Ai


class MotionModelConstantVelocity:
    """motion = previous^-1 * current (motion_model_constant_velocity.hpp:17-19)."""

    def __init__(self, dim=3):
        self.dim = dim
        self.clear()

    def clear(self):  # :42-46
        self._motion = identity(self.dim)
        self._robot_in_local_map = identity(self.dim)
        self._robot_in_local_map_previous = identity(self.dim)

    def estimate(self):
        return self._motion

    def compute(self):
        self._motion = compose(inverse(self._robot_in_local_map_previous), self._robot_in_local_map)

    def set_robot_in_local_map(self, robot_in_local_map):  # :21-25
        self._robot_in_local_map_previous = self._robot_in_local_map
        self._robot_in_local_map = np.asarray(robot_in_local_map, np.float32)

    def shift_tracker_estimate(self, estimate):  # :28-35
        estimate = np.asarray(estimate, np.float32)
        self._robot_in_local_map_previous = compose(
            estimate, compose(inverse(self._robot_in_local_map_previous), self._robot_in_local_map))
        self._robot_in_local_map = estimate

    def set_robot_in_local_map_previous(self, robot_in_local_map):  # :37-40
        self._robot_in_local_map_previous = np.asarray(robot_in_local_map, np.float32)


class TrackerEstimateBuffer:
    """RawDataPreprocessorTrackerEstimate_: rolling buffer of tracker poses (param number_of_poses_to_keep = 5)."""

    def __init__(self, dim=3, number_of_poses_to_keep=5):
        self.dim = dim
        self.number_of_poses_to_keep = number_of_poses_to_keep
        self._robot_in_local_map = identity(dim)
        self.estimates = []

    def set_robot_in_local_map(self, robot_in_local_map):  # :54-56
        self._robot_in_local_map = np.asarray(robot_in_local_map, np.float32)

    def compute(self):  # :30-44
        if len(self.estimates) == self.number_of_poses_to_keep:
            self.estimates.pop(0)
        self.estimates.append(self._robot_in_local_map)
        return list(self.estimates)

    def set_coordinate_frame_origin(self, origin):  # :60-68
        into_new = inverse(origin)
        self.estimates = [compose(into_new, e) for e in self.estimates]


class AlignerSliceMotionModel:
    """Prior slice whose measurement is the inverse constant-velocity motion (aligner_slice_motion_model.hpp)."""

    def __init__(self, aligner, motion_model=None, information_diag=None):
        self.aligner = aligner
        self.motion_model = motion_model
        cfg = abi.default_slice_config(aligner.variable_kind)
        cfg.kind = abi.SLICE_PRIOR
        cfg.finder = abi.FINDER_NONE
        cfg.prior_sets_initial_guess = 1  # init() calls aligner->setMovingInFixed(_motion_inverse), :69-70
        if information_diag is not None:
            for i, v in enumerate(information_diag):
                cfg.prior_information_diag[i] = v
        self.slice_idx = aligner.add_slice(cfg)
        self._fixed_slice = None

    def set_fixed(self, pose_buffer):
        """The fixed slice is the tracker pose deque (StdDequeEigenIsometry3f)."""
        self._fixed_slice = pose_buffer

    def init(self):
        if self.motion_model is None:
            raise RuntimeError("AlignerSliceMotionModel_::init|ERROR: no motion model is set")  # :45-47
        if self._fixed_slice is None:
            raise RuntimeError("AlignerSliceMotionModel_::init|ERROR: no fixed pose set")  # :48-50
        if len(self._fixed_slice):  # :55-62
            self.motion_model.set_robot_in_local_map(self._fixed_slice[-1])
            if len(self._fixed_slice) > 1:
                self.motion_model.set_robot_in_local_map_previous(self._fixed_slice[-2])
            self.motion_model.compute()
        motion_inverse = inverse(self.motion_model.estimate())  # :69
        self.aligner.set_prior_measurement(self.slice_idx, motion_inverse)
        return motion_inverse


class AlignerSliceOdomPrior:
    """AlignerSliceOdom2DPrior / 3DPrior: measurement = fixed^-1 * moving after the second call."""

    def __init__(self, aligner, information_diag=None):
        self.aligner = aligner
        cfg = abi.default_slice_config(aligner.variable_kind)
        cfg.kind = abi.SLICE_PRIOR
        cfg.finder = abi.FINDER_NONE
        cfg.prior_sets_initial_guess = 1
        if information_diag is not None:
            for i, v in enumerate(information_diag):
                cfg.prior_information_diag[i] = v
        self.slice_idx = aligner.add_slice(cfg)
        self._count = 0
        self._fixed = None
        self._moving = None

    def set_fixed(self, odom_pose):
        self._fixed = np.asarray(odom_pose, np.float32)

    def set_moving(self, odom_pose):
        self._moving = np.asarray(odom_pose, np.float32)

    def init(self):
        if self._fixed is None:
            raise RuntimeError("AlignerSliceProcessor_::factor| no fixed")  # aligner_slice_processor_prior_impl.cpp:16
        if self._moving is None:
            raise RuntimeError("AlignerSliceProcessor_::factor| no moving")  # :20
        delta = identity(self.aligner.dim)
        if self._count > 1:  # aligner_slice_odometry_prior.cpp:8-10,25-27
            delta = compose(inverse(self._fixed), self._moving)
        self.aligner.set_prior_measurement(self.slice_idx, delta)
        self._count += 1
        return delta
