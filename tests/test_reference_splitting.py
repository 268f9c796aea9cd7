"""The reference's 14 splitting-criterion tests (T/test_local_map_splitting_criterion.cpp:12-376) restated against
srrg2_slam_interfaces_amd/splitting.py, plus the visibility criterion's branches (local_map_splitting_criterion_
visibility.hpp:33-98), which the reference does not test."""
import math

import numpy as np
import pytest

from srrg2_slam_interfaces_amd import splitting as sp


class Slam:
    """what MultiGraphSLAM{2,3}D exposes to a criterion"""

    def __init__(self, dim):
        self.T = np.eye(3) if dim == 2 else np.eye(4)[:3]
        self.points, self.reloc, self.stats, self.reloc_stats = None, False, [], []

    def set_robot_in_local_map(self, T):
        self.T = np.array(T, dtype=np.float64)

    def robot_in_local_map(self):
        return self.T

    def current_local_map_points(self):
        return self.points

    def relocalized(self):
        return self.reloc

    def tracker_iteration_stats(self):
        return self.stats

    def relocalizer_iteration_stats(self):
        return self.reloc_stats


def axis_rot(axis, a):
    c, s = math.cos(a), math.sin(a)
    R = np.eye(3)
    i, j = (axis + 1) % 3, (axis + 2) % 3
    R[i, i], R[i, j], R[j, i], R[j, j] = c, -s, s, c
    return R


def rotate(T, R):  # Isometry::rotate: T.linear() *= R
    T = T.copy()
    T[:, :3] = T[:, :3] @ R
    return T


def rotate2(T, a):
    T = T.copy()
    c, s = math.cos(a), math.sin(a)
    T[:2, :2] = T[:2, :2] @ np.array([[c, -s], [s, c]])
    return T


@pytest.mark.parametrize("dim", [2, 3])
def test_distance_translation_1d(dim):  # :12-62
    crit = sp.LocalMapSplittingCriterionDistance(5.0)
    slam = Slam(dim)
    crit.set_slam_algorithm(slam)
    T = slam.robot_in_local_map().copy()
    for i in range(10):
        slam.set_robot_in_local_map(T)
        crit.compute()
        assert crit.has_to_split() == (i > 5)
        T[0, -1] += 1.0


@pytest.mark.parametrize("sign", [-1.0, 1.0])
def test_rotation_2d(sign):  # :64-112 (the reference's "Negative" case also steps positively)
    crit = sp.LocalMapSplittingCriterionRotation(math.pi / 6)
    slam = Slam(2)
    crit.set_slam_algorithm(slam)
    step = sign * crit.param_local_map_angle_distance_radians / 10
    T = slam.robot_in_local_map().copy()
    for _ in range(10):
        slam.set_robot_in_local_map(T)
        crit.compute()
        assert not crit.has_to_split()
        T = rotate2(T, step)
    T = rotate2(T, step)
    slam.set_robot_in_local_map(T)
    crit.compute()
    assert crit.has_to_split()


@pytest.mark.parametrize("axis", [0, 1, 2])
@pytest.mark.parametrize("sign", [1.0, -1.0])
def test_rotation_3d_single_axis(axis, sign):  # :114-187, :218-291
    crit = sp.LocalMapSplittingCriterionRotation(math.pi / 6)
    slam = Slam(3)
    crit.set_slam_algorithm(slam)
    step = sign * crit.param_local_map_angle_distance_radians / 10
    T = slam.robot_in_local_map().copy()
    for _ in range(10):
        slam.set_robot_in_local_map(T)
        crit.compute()
        assert not crit.has_to_split()
        T = rotate(T, axis_rot(axis, step))
    T = rotate(T, axis_rot(axis, step))
    slam.set_robot_in_local_map(T)
    crit.compute()
    assert crit.has_to_split()


@pytest.mark.parametrize("signs", [(1, 1, -1), (-1, -1, 1)])
@pytest.mark.parametrize("cls", ["rotation", "viewpoint"])
def test_rotation_3d_combined(signs, cls):  # :189-216, :293-320, :322-350
    if cls == "rotation":
        crit = sp.LocalMapSplittingCriterionRotation(math.pi / 6)
    else:
        if signs != (-1, -1, 1):
            pytest.skip("the reference tests the viewpoint criterion with one sign pattern")
        crit = sp.LocalMapSplittingCriterionViewpoint(1.0, math.pi / 6)
    slam = Slam(3)
    crit.set_slam_algorithm(slam)
    step = crit.param_local_map_angle_distance_radians / 10
    R = axis_rot(0, signs[0] * step) @ axis_rot(1, signs[1] * step) @ axis_rot(2, signs[2] * step)
    T = slam.robot_in_local_map().copy()
    for _ in range(6):
        slam.set_robot_in_local_map(T)
        crit.compute()
        assert not crit.has_to_split()
        T = rotate(T, R)
    T = rotate(T, R)
    slam.set_robot_in_local_map(T)
    crit.compute()
    assert crit.has_to_split()


def test_viewpoint_translation_positive_x():  # :352-376
    crit = sp.LocalMapSplittingCriterionViewpoint(1.0, math.pi / 6)
    slam = Slam(3)
    crit.set_slam_algorithm(slam)
    step = crit.param_local_map_distance / 10
    T = slam.robot_in_local_map().copy()
    for _ in range(10):
        slam.set_robot_in_local_map(T)
        crit.compute()
        assert not crit.has_to_split()
        T[0, 3] = np.float32(T[0, 3]) + np.float32(step)
    T[0, 3] = np.float32(T[0, 3]) + np.float32(step)
    slam.set_robot_in_local_map(T)
    crit.compute()
    assert crit.has_to_split()


def test_criteria_throw_without_slam_algorithm():
    for crit in (sp.LocalMapSplittingCriterionDistance(), sp.LocalMapSplittingCriterionRotation(),
                 sp.LocalMapSplittingCriterionViewpoint(), sp.LocalMapSplittingCriterionVisibility()):
        with pytest.raises(RuntimeError):
            crit.compute()


def test_visibility_branches():
    crit = sp.LocalMapSplittingCriterionVisibility(1000, 0.1)
    slam = Slam(3)
    crit.set_slam_algorithm(slam)
    crit.compute()  # no local map: no decision
    assert not crit.has_to_split()
    slam.points, slam.stats = 500, [{"num_inliers": 400}]
    crit.compute()
    assert not crit.has_to_split()
    slam.points = 1200  # grew beyond the limit
    crit.compute()
    assert crit.has_to_split()
    slam.points, slam.stats = 600, [{"num_inliers": 30}]  # tracked ratio 5 % < 10 %
    crit.compute()
    assert crit.has_to_split()
    slam.points, slam.stats, slam.reloc, slam.reloc_stats = 600, [{"num_inliers": 30}], True, [{"num_inliers": 300}]
    crit.compute()  # relocalized: the relocalizer's statistics count
    assert not crit.has_to_split()
    slam.reloc, slam.stats = False, []
    crit.compute()  # no statistics: no decision
    assert not crit.has_to_split()


def test_euler_angles_roundtrip():
    rng = np.random.default_rng(0)
    for _ in range(100):
        a = rng.uniform(-1.2, 1.2, 3)
        R = axis_rot(0, a[0]) @ axis_rot(1, a[1]) @ axis_rot(2, a[2])
        e = sp.euler_angles_012(R)
        R2 = axis_rot(0, e[0]) @ axis_rot(1, e[1]) @ axis_rot(2, e[2])
        assert np.allclose(R, R2, atol=1e-12)
        assert 0.0 <= e[0] <= math.pi + 1e-12
