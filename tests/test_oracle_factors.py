"""CPU: the oracle's factor arithmetic against finite differences of an independent float64 cost.

For fixed correspondences the Gauss-Newton quantities must satisfy  b = J^T e  and  H = J^T J  where
J = d e(X [+] dx) / d dx at 0.  e() is re-implemented here in numpy float64 and differentiated numerically
using the oracle's own box_plus for the perturbation (which is itself checked in test_oracle_math.py)."""
import numpy as np
import pytest

from helpers import cue_config, prior_config, setup_pair
from srrg2_slam_interfaces_amd import _abi as abi
from srrg2_slam_interfaces_amd import synthetic as syn


def _acc_to_system(acc, k, D):
    H = np.zeros((D, D))
    b = np.zeros(D)
    for a in range(D):
        for c in range(a, D):
            idx = a * 6 - (a * (a - 1)) // 2 + (c - a)
            H[a, c] = H[c, a] = float(acc[idx]) * 2.0 ** (-k)
        b[a] = float(acc[21 + a]) * 2.0 ** (-k)
    return H, b


def _residuals(kind, slice_kind, X, data, corr):
    X = np.asarray(X, np.float64)
    p = data["moving"][corr["moving_idx"]].astype(np.float64)
    f = data["fixed"][corr["fixed_idx"]].astype(np.float64)
    if kind == abi.SE2_RIGHT:
        q = p @ X[:2, :2].T + X[:2, 2]
    else:
        q = p @ X[:, :3].T + X[:, 3]
    if slice_kind == abi.SLICE_P2PLANE:
        n = data["fixed_normals"][corr["fixed_idx"]].astype(np.float64)
        return np.sum(n * (q - f), axis=1)
    return (q - f).reshape(-1)


@pytest.mark.parametrize("kind,slice_kind", [
    (abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE), (abi.SE3_QUAT_RIGHT, abi.SLICE_P2P),
    (abi.SE3_EULER_RIGHT, abi.SLICE_P2PLANE), (abi.SE3_EULER_RIGHT, abi.SLICE_P2P),
    (abi.SE2_RIGHT, abi.SLICE_P2P), (abi.SE2_RIGHT, abi.SLICE_P2PLANE)])
def test_H_b_match_finite_differences(oracle, kind, slice_kind):
    if kind == abi.SE2_RIGHT:
        d = syn.scan_pair_2d(beams=400, sigma=0.005)
        guess = syn.se2(0.05, -0.02, 0.03).astype(np.float32)
        gate, D = 0.5, 3
    else:
        d = syn.cloud_pair_3d(n=1500, seed=21, noise_sigma=0.003)
        guess = syn.se3(np.array([0.02, 0.01, -0.01]), np.array([0.01, -0.02, 0.015])).astype(np.float32)
        gate, D = 0.3, 6
    al = oracle.OracleAligner(kind)
    setup_pair(al, d, cue_config(kind, slice_kind, gate), guess)
    acc, k = al.linearize_once(0)
    corr = al.correspondences(0)
    assert len(corr) > 200
    H, b = _acc_to_system(acc, k, D)
    e0 = _residuals(kind, slice_kind, guess, d, corr)
    J = np.zeros((e0.size, D))
    eps = 1e-4
    for a in range(D):
        dx = np.zeros(D)
        dx[a] = eps
        ep = _residuals(kind, slice_kind, oracle.box_plus(kind, guess, dx), d, corr)
        em = _residuals(kind, slice_kind, oracle.box_plus(kind, guess, -dx), d, corr)
        J[:, a] = (ep - em) / (2 * eps)
    scale_H = np.abs(J.T @ J).max()
    assert np.abs(H - J.T @ J).max() / scale_H < 2e-3  # float32 perturbation noise dominates
    assert np.abs(b - J.T @ e0).max() / max(np.abs(J.T @ e0).max(), 1e-9) < 2e-3
    # statistics: no robustifier -> every correspondence is an inlier, chi = sum e^2
    assert acc[29] == len(corr) and acc[30] == 0 and acc[31] == len(corr)
    assert abs(float(acc[27]) * 2.0 ** (-k) - float(np.sum(e0 ** 2))) / float(np.sum(e0 ** 2)) < 1e-4


@pytest.mark.parametrize("rob", [abi.ROBUST_CLAMP, abi.ROBUST_SATURATED, abi.ROBUST_CAUCHY])
def test_robustifier_weights(oracle, rob):
    kind = abi.SE3_QUAT_RIGHT
    d = syn.cloud_pair_3d(n=1500, seed=22, noise_sigma=0.01)
    thr = 1e-4
    al = oracle.OracleAligner(kind)
    setup_pair(al, d, cue_config(kind, abi.SLICE_P2PLANE, 0.3, rob, thr))
    acc, k = al.linearize_once(0)
    corr = al.correspondences(0)
    fs = al.factor_status(0)
    e = _residuals(kind, abi.SLICE_P2PLANE, syn.identity(3), d, corr)
    chi = (e.astype(np.float32) ** 2).astype(np.float64)
    out = chi >= thr
    assert 0 < out.sum() < len(corr)
    assert np.array_equal(fs == abi.FACTOR_KERNELIZED, out)
    assert acc[29] == (~out).sum() and acc[30] == out.sum()
    w = np.ones_like(chi)
    if rob == abi.ROBUST_CLAMP:
        w[out] = 0.0
    elif rob == abi.ROBUST_SATURATED:
        w[out] = thr / chi[out]
    else:
        w[out] = 1.0 / (1.0 + chi[out] / thr)
    # translational block of H for point-to-plane: sum w n n^T
    n = d["fixed_normals"][corr["fixed_idx"]].astype(np.float64)
    Htt = (n * w[:, None]).T @ n
    H, _ = _acc_to_system(acc, k, 6)
    assert np.allclose(H[:3, :3], Htt, rtol=1e-4, atol=1e-6)
    assert abs(float(acc[28]) * 2.0 ** (-k) - chi[out].sum()) / chi[out].sum() < 1e-4


def test_prior_factor_is_zero_at_its_measurement_and_pulls_towards_it(oracle):
    for kind, Z, X0 in ((abi.SE3_QUAT_RIGHT, syn.se3(np.array([0.3, -0.2, 0.1]), np.deg2rad([20., -30., 45.])),
                         syn.se3(np.array([0.25, -0.1, 0.0]), np.deg2rad([15., -25., 40.]))),
                        (abi.SE2_RIGHT, syn.se2(0.4, -0.1, 0.7), syn.se2(0.3, 0.0, 0.6))):
        al = oracle.OracleAligner(kind)
        al.set_params(max_iterations=8, min_num_inliers=0)
        pi = al.add_slice(prior_config(kind, sets_guess=0))
        al.set_prior_measurement(pi, Z.astype(np.float32))
        al.set_moving_in_fixed(X0.astype(np.float32))
        assert al.compute() == abi.SUCCESS
        st = al.iteration_stats()
        assert st[0]["chi_inliers"] > 1e-3 and st[-1]["chi_inliers"] < 1e-9  # Gauss-Newton converges onto Z
        assert np.allclose(al.moving_in_fixed(), Z, atol=2e-6)
        assert all(s["num_inliers"] == 1 and s["num_correspondences"] == 1 for s in st)
