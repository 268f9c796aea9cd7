"""CPU: the oracle's deterministic math against libm / numpy (the oracle is the parity anchor, so it is
itself checked against independent implementations first)."""
import math

import numpy as np

from srrg2_slam_interfaces_amd import _abi as abi
from srrg2_slam_interfaces_amd import synthetic as syn


def test_sincos_atan2_match_libm(oracle):
    xs = np.concatenate([np.linspace(-12, 12, 4001), np.array([0.0, 1e-9, -1e-9, math.pi, -math.pi / 2])])
    for x in xs:
        s, c = oracle.sincos(float(x))
        assert abs(s - math.sin(x)) < 4e-16 and abs(c - math.cos(x)) < 4e-16
    for a in np.linspace(-math.pi, math.pi, 2001):
        for r in (1e-3, 1.0, 37.0):
            y, x = r * math.sin(a), r * math.cos(a)
            assert abs(oracle.atan2(y, x) - math.atan2(y, x)) < 1e-15
    assert oracle.atan2(0.0, 0.0) == 0.0
    assert abs(oracle.atan2(1.0, 0.0) - math.pi / 2) < 1e-15
    assert abs(oracle.atan2(-1.0, 0.0) + math.pi / 2) < 1e-15


def test_se3_se2_group_ops(oracle):
    A = syn.se3(np.array([0.3, -1.2, 2.0]), np.array([0.4, -0.2, 1.1])).astype(np.float32)
    B = syn.se3(np.array([-0.7, 0.1, 0.5]), np.array([-1.0, 0.3, 0.2])).astype(np.float32)
    C = oracle.se3_compose(A, B)
    assert np.allclose(C, syn.se3_mul(A.astype(np.float64), B.astype(np.float64)), atol=1e-6)
    I = oracle.se3_compose(A, oracle.se3_inverse(A))
    assert np.allclose(I, np.eye(3, 4), atol=1e-6)
    A2 = syn.se2(0.4, -0.3, 0.9).astype(np.float32)
    B2 = syn.se2(-1.0, 2.0, -2.2).astype(np.float32)
    assert np.allclose(oracle.se2_compose(A2, B2), A2.astype(np.float64) @ B2.astype(np.float64), atol=1e-6)
    assert np.allclose(oracle.se2_compose(A2, oracle.se2_inverse(A2)), np.eye(3), atol=1e-6)


def test_v2t_t2v_roundtrip_and_box_plus(oracle):
    # quaternion flavour: v[3:6] = imaginary part of the unit quaternion
    ang = 0.7
    axis = np.array([1.0, 2.0, -1.5])
    axis /= np.linalg.norm(axis)
    v = np.concatenate([[0.1, -0.2, 0.3], np.sin(ang / 2) * axis])
    R, t = oracle.se3_v2t(abi.SE3_QUAT_RIGHT, v)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R_true = np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * K @ K
    assert np.allclose(R, R_true, atol=1e-14) and np.allclose(t, v[:3])
    T = np.zeros((3, 4), np.float32)
    T[:, :3], T[:, 3] = R, t
    assert np.allclose(oracle.se3_t2v_quat(T), v, atol=1e-6)
    # Euler flavour: R = Rx Ry Rz
    ve = np.array([0.0, 0.0, 0.0, 0.3, -0.4, 1.2])
    Re, _ = oracle.se3_v2t(abi.SE3_EULER_RIGHT, ve)
    assert np.allclose(Re, syn.rpy_to_R(0.3, -0.4, 1.2), atol=1e-14)
    # |q| >= 1 -> identity rotation
    Rb, _ = oracle.se3_v2t(abi.SE3_QUAT_RIGHT, np.array([0, 0, 0, 1.0, 0.5, 0.0]))
    assert np.array_equal(Rb, np.eye(3))
    # box_plus = X * v2t(dx), rounded once to float32
    X = syn.se3(np.array([1.0, 2.0, 3.0]), np.array([0.2, 0.1, -0.3])).astype(np.float32)
    Xn = oracle.box_plus(abi.SE3_QUAT_RIGHT, X, v)
    D = np.zeros((3, 4))
    D[:, :3], D[:, 3] = R, t
    assert np.allclose(Xn, syn.se3_mul(X.astype(np.float64), D), atol=1e-6)
    X2 = syn.se2(0.5, -0.5, 0.3).astype(np.float32)
    X2n = oracle.box_plus(abi.SE2_RIGHT, X2, np.array([0.1, 0.2, -0.4]))
    assert np.allclose(X2n, X2.astype(np.float64) @ syn.se2(0.1, 0.2, -0.4), atol=1e-6)
    assert np.allclose(oracle.se2_t2v(X2n)[2], 0.3 - 0.4, atol=1e-6)


def test_fix_transform_orthonormalises(oracle):
    X = syn.se3(np.array([1.0, 2.0, 3.0]), np.array([0.2, 0.1, -0.3])).astype(np.float32)
    Xd = X.copy()
    Xd[:, :3] *= np.float32(1.001)
    Xd[0, 1] += np.float32(1e-3)
    F = oracle.fix_transform(abi.SE3_QUAT_RIGHT, Xd)
    R = F[:, :3].astype(np.float64)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-6) and abs(np.linalg.det(R) - 1) < 1e-6
    assert np.array_equal(F[:, 3], Xd[:, 3])
    F2 = oracle.fix_transform(abi.SE2_RIGHT, (syn.se2(1, 2, 0.5) * 1.01).astype(np.float32))
    assert abs(F2[0, 0] ** 2 + F2[1, 0] ** 2 - 1) < 1e-6 and F2[0, 1] == -F2[1, 0] and F2[1, 1] == F2[0, 0]


def test_dense_solve(oracle):
    rng = np.random.default_rng(3)
    for D in (3, 6):
        A = rng.normal(size=(20, D))
        H = A.T @ A
        b = rng.normal(size=D)
        rc, dx = oracle.solve(H, b)
        assert rc == 0 and np.allclose(dx, np.linalg.solve(H, -b), rtol=1e-10, atol=1e-12)
        H[0, :] = 0
        H[:, 0] = 0
        rc, _ = oracle.solve(H, b)
        assert rc == 1  # not positive definite -> solver status != Success


def test_fixed_point_exponent(oracle):
    f = oracle.lib().o_fixed_point_exponent
    assert f(1, 1.0) == 50  # clamped
    assert f(100000, 600.0) == 62 - 17 - 10
    assert f(1 << 20, 1.0) == 62 - 20 - 0
    assert f(3, 2.0) == 49  # per-term cap: 2^k * bound <= 2^50
