"""Local-map splitting criteria (host logic; S/mapping/local_map_splitting_criterions/*.hpp): they decide when the
SLAM system opens a new local map, i.e. when ``GraphSLAMLifecycle.make_new_map`` is called.

  LocalMapSplittingCriterionDistance     local_map_splitting_criterion_translation.hpp:24-33
  LocalMapSplittingCriterionRotation     local_map_splitting_criterion_rotation.hpp:29-58
  LocalMapSplittingCriterionViewpoint    local_map_splitting_criterion_viewpoint.hpp:34-54
  LocalMapSplittingCriterionVisibility   local_map_splitting_criterion_visibility.hpp:33-98

The criteria talk to a *slam algorithm* object exposing what they read in the reference: ``robot_in_local_map()``
(3x3 / 3x4 array), and for the visibility criterion ``current_local_map_points()`` (int or None), ``relocalized()``,
``tracker_iteration_stats()`` / ``relocalizer_iteration_stats()`` (lists of dicts with ``num_inliers``).
The 3-D rotation criterion uses Eigen's ``eulerAngles(0, 1, 2)`` (un-vendored dependency of the reference, Eigen 3.3
``Geometry/EulerAngles.h``), restated in ``euler_angles_012``.
"""
import math

import numpy as np


def euler_angles_012(R):
    """Eigen 3.3 MatrixBase::eulerAngles(0, 1, 2): R = Rx(a0) Ry(a1) Rz(a2), a0 in [0, pi]."""
    i, j, k = 0, 1, 2  # a0 = 0, a1 = 1 -> "even" permutation
    r0 = math.atan2(R[j, k], R[k, k])
    c2 = math.hypot(R[i, i], R[i, j])
    if r0 > 0.0:
        r0 = r0 - math.pi
        r1 = math.atan2(-R[i, k], -c2)
    else:
        r1 = math.atan2(-R[i, k], c2)
    s1, c1 = math.sin(r0), math.cos(r0)
    r2 = math.atan2(s1 * R[k, i] - c1 * R[j, i], c1 * R[j, j] - s1 * R[k, j])
    return np.array([-r0, -r1, -r2])


class _Criterion:
    def __init__(self):
        self._slam = None
        self._has_to_split = False

    def set_slam_algorithm(self, slam):
        self._slam = slam

    def has_to_split(self):
        return self._has_to_split

    def _pose(self, who):
        if self._slam is None:
            raise RuntimeError("%s|SLAM algorithm not set" % who)
        return np.asarray(self._slam.robot_in_local_map(), dtype=np.float64)


class LocalMapSplittingCriterionDistance(_Criterion):
    def __init__(self, local_map_distance=1.0):
        super().__init__()
        self.param_local_map_distance = float(local_map_distance)

    def compute(self):
        self._has_to_split = False
        T = self._pose("LocalMapSplittingCriterionDistance")
        t = T[:2, 2] if T.shape == (3, 3) else T[:, 3]
        if float(np.float32(np.linalg.norm(t.astype(np.float32)))) > self.param_local_map_distance:
            self._has_to_split = True


class LocalMapSplittingCriterionRotation(_Criterion):
    def __init__(self, local_map_angle_distance_radians=0.5):
        super().__init__()
        self.param_local_map_angle_distance_radians = float(local_map_angle_distance_radians)

    @staticmethod
    def _delta(T):
        if T.shape == (3, 3):  # SE(2): |atan2(r10, r00)|, :45-47
            return abs(math.atan2(T[1, 0], T[0, 0]))
        d = euler_angles_012(T[:, :3])  # :48-57 (valid for small angular changes, as the reference notes)
        d = np.minimum(np.abs(d), math.pi + 1e-5 - np.abs(d))
        return float(np.linalg.norm(d))

    def compute(self):
        self._has_to_split = False
        T = self._pose("LocalMapSplittingCriterionRotation")
        if self._delta(T) > self.param_local_map_angle_distance_radians:
            self._has_to_split = True


class LocalMapSplittingCriterionViewpoint(_Criterion):
    """translation OR rotation (aggregation, :39-53)"""

    def __init__(self, local_map_distance=1.0, local_map_angle_distance_radians=0.5):
        super().__init__()
        self.param_local_map_distance = float(local_map_distance)
        self.param_local_map_angle_distance_radians = float(local_map_angle_distance_radians)
        self._translation = LocalMapSplittingCriterionDistance()
        self._rotation = LocalMapSplittingCriterionRotation()

    def compute(self):
        self._has_to_split = False
        self._pose("LocalMapSplittingCriterionViewpoint")
        self._translation.set_slam_algorithm(self._slam)
        self._translation.param_local_map_distance = self.param_local_map_distance
        self._translation.compute()
        if self._translation.has_to_split():
            self._has_to_split = True
            return
        self._rotation.set_slam_algorithm(self._slam)
        self._rotation.param_local_map_angle_distance_radians = self.param_local_map_angle_distance_radians
        self._rotation.compute()
        if self._rotation.has_to_split():
            self._has_to_split = True


class LocalMapSplittingCriterionVisibility(_Criterion):
    def __init__(self, maximum_number_of_points=1000, minimum_tracked_point_ratio=0.1):
        super().__init__()
        self.param_maximum_number_of_points = float(maximum_number_of_points)
        self.param_minimum_tracked_point_ratio = float(minimum_tracked_point_ratio)
        self._previous_number_of_points = 0

    def compute(self):
        self._has_to_split = False
        if self._slam is None:
            raise RuntimeError("LocalMapSplittingCriterionVisbility|SLAM algorithm not set")
        n = self._slam.current_local_map_points()
        if n is None:  # no local map available: no decision (:45-51)
            self._previous_number_of_points = 0
            return
        if self._previous_number_of_points != 0:
            delta = n - self._previous_number_of_points
            if delta != 0 and n > self.param_maximum_number_of_points:  # :59-65
                self._has_to_split = True
                self._previous_number_of_points = 0
                return
        self._previous_number_of_points = n
        stats = (self._slam.relocalizer_iteration_stats() if self._slam.relocalized()
                 else self._slam.tracker_iteration_stats())  # :72-84
        if not stats:
            return
        ratio = float(stats[-1]["num_inliers"]) / n if n else float("inf")
        if ratio < self.param_minimum_tracked_point_ratio:  # :92-97
            self._has_to_split = True
            self._previous_number_of_points = 0
