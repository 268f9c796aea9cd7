"""Golden vectors (tests/golden/icp_golden.npz, made by tests/golden/make_golden.py from an independent
numpy restatement) pin one ICP iteration: correspondence indices + responses bit-exact, H / b / dx / X within
tolerance.  CPU leg: the oracle; gpu leg: the HIP library alone (no oracle involved)."""
import os

import numpy as np
import pytest

from helpers import cue_config, setup_pair
from srrg2_slam_interfaces_amd import _abi as abi
from srrg2_slam_interfaces_amd import synthetic as syn

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "icp_golden.npz"))
BACKENDS = ["oracle", pytest.param("hip", marks=pytest.mark.gpu)]


def _aligner(backend, oracle, kind):
    if backend == "oracle":
        return oracle.OracleAligner(kind)
    import srrg2_slam_interfaces_amd as pkg

    return pkg.MultiAligner(kind)


def _check_corr(c, idx, d2):
    sel = idx >= 0
    assert np.array_equal(c["moving_idx"], np.nonzero(sel)[0].astype(np.int32))
    assert np.array_equal(c["fixed_idx"], idx[sel])
    assert c["response"].tobytes() == d2[sel].tobytes()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("gi", [0, 1])
@pytest.mark.parametrize("name,slice_kind,rob", [("plane", abi.SLICE_P2PLANE, abi.ROBUST_CAUCHY),
                                                  ("p2p", abi.SLICE_P2P, abi.ROBUST_NONE)])
def test_one_iteration_se3(backend, oracle, gi, name, slice_kind, rob):
    d = syn.cloud_pair_3d(n=2000, seed=123)
    al = _aligner(backend, oracle, abi.SE3_QUAT_RIGHT)
    al.set_params(max_iterations=1)
    setup_pair(al, d, cue_config(abi.SE3_QUAT_RIGHT, slice_kind, 0.25, rob, 0.05), GOLD["a%d_guess" % gi])
    assert al.compute() == abi.SUCCESS
    _check_corr(al.correspondences(0), GOLD["a%d_idx" % gi], GOLD["a%d_d2" % gi])
    X_gold = GOLD["a%d_%s_X" % (gi, name)]
    assert np.max(np.abs(al.moving_in_fixed() - X_gold)) <= 1e-5  # increments within 1e-5 (north_star)
    if backend == "oracle":
        H, b, dx = al.last_system()
        Hg, bg, dxg = GOLD["a%d_%s_H" % (gi, name)], GOLD["a%d_%s_b" % (gi, name)], GOLD["a%d_%s_dx" % (gi, name)]
        assert np.max(np.abs(H - Hg)) / np.max(np.abs(Hg)) < 1e-5  # float32 J entries vs float64 J
        assert np.max(np.abs(b - bg)) / np.max(np.abs(bg)) < 1e-4
        assert np.max(np.abs(dx - dxg)) < 1e-5


@pytest.mark.parametrize("backend", BACKENDS)
def test_one_iteration_c1_se2(backend, oracle):
    d = syn.scan_pair_2d(beams=1000)
    al = _aligner(backend, oracle, abi.SE2_RIGHT)
    al.set_params(max_iterations=1)
    setup_pair(al, d, cue_config(abi.SE2_RIGHT, abi.SLICE_P2P, 0.5))
    assert al.compute() == abi.SUCCESS
    _check_corr(al.correspondences(0), GOLD["c_idx"], GOLD["c_d2"])
    assert np.max(np.abs(al.moving_in_fixed() - GOLD["c_X"])) <= 1e-5
    if backend == "oracle":
        H, b, dx = al.last_system()
        assert np.max(np.abs(H - GOLD["c_H"])) / np.max(np.abs(GOLD["c_H"])) < 1e-5
        assert np.max(np.abs(dx - GOLD["c_dx"])) < 1e-5


# ---- second derivation: finite-difference Jacobians on the golden side (tests/golden/make_golden_fd.py) --------------------
GOLD_FD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "icp_golden_fd.npz"))


@pytest.mark.parametrize("backend", BACKENDS)
def test_reprojection_factor_against_finite_differences(backend, oracle):
    """projective finder + pinhole reprojection factor (the second slice of BASELINE config C3): correspondences and
    responses bit-exact against the restated finder, H / b / dx / X against a Gauss-Newton step whose Jacobians are
    central finite differences of the residual function"""
    from helpers import projective_config

    d = syn.rgbd_pair(rows=60, cols=80, seed=3100)
    al = _aligner(backend, oracle, abi.SE3_QUAT_RIGHT)
    al.set_params(max_iterations=1, min_num_inliers=10)
    setup_pair(al, d, projective_config(abi.SE3_QUAT_RIGHT, abi.SLICE_REPROJECTION, d, gate=0.05), GOLD_FD["r_guess"])
    assert al.compute() == abi.SUCCESS
    c = al.correspondences(0)
    match = GOLD_FD["r_match"]
    sel = match >= 0
    assert np.array_equal(c["moving_idx"], np.nonzero(sel)[0].astype(np.int32))
    assert np.array_equal(c["fixed_idx"], match[sel])
    assert c["response"].tobytes() == GOLD_FD["r_resp"][sel].tobytes()
    assert al.iteration_stats()[0]["num_inliers"] == int(GOLD_FD["r_n"])
    assert np.max(np.abs(al.moving_in_fixed() - GOLD_FD["r_X"])) <= 1e-5
    if backend == "oracle":
        H, b, dx = al.last_system()
        assert np.max(np.abs(H - GOLD_FD["r_H"])) / np.max(np.abs(GOLD_FD["r_H"])) < 1e-4
        assert np.max(np.abs(b - GOLD_FD["r_b"])) / np.max(np.abs(GOLD_FD["r_b"])) < 1e-3
        assert np.max(np.abs(dx - GOLD_FD["r_dx"])) < 1e-5


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("gi", [0, 1])
def test_se3_euler_box_plus_against_finite_differences(backend, oracle, gi):
    """MultiAligner3D (VariableSE3EulerRightAD): X <- X * [Rx Ry Rz | t] with a finite-difference Jacobian"""
    d = syn.cloud_pair_3d(n=1500, seed=321)
    al = _aligner(backend, oracle, abi.SE3_EULER_RIGHT)
    al.set_params(max_iterations=1)
    setup_pair(al, d, cue_config(abi.SE3_EULER_RIGHT, abi.SLICE_P2P, 0.25), GOLD_FD["e%d_guess" % gi])
    assert al.compute() == abi.SUCCESS
    _check_corr(al.correspondences(0), GOLD_FD["e%d_idx" % gi], GOLD_FD["e%d_d2" % gi])
    assert np.max(np.abs(al.moving_in_fixed() - GOLD_FD["e%d_X" % gi])) <= 1e-5
    if backend == "oracle":
        H, b, dx = al.last_system()
        assert np.max(np.abs(H - GOLD_FD["e%d_H" % gi])) / np.max(np.abs(GOLD_FD["e%d_H" % gi])) < 1e-5
        assert np.max(np.abs(dx - GOLD_FD["e%d_dx" % gi])) < 1e-5
