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


def _H_of_last_iteration(al, backend, data, guess, with_normals=True):
    """H = sum w J^T J (+ priors) of the last Gauss-Newton iteration.  The oracle exposes its last system; the product
    exposes H through the batch result record (srrg2_batch_result::information, float32): the same alignment run once
    more as a batch of one."""
    if backend == "oracle":
        return al.last_system()[0]
    res = al.compute_batch([data["moving"]], [guess], [data["moving_normals"]] if with_normals else None)
    assert res[0]["status"] == al.status()
    return res[0]["information"].astype(np.float64)


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
    Hg = GOLD["a%d_%s_H" % (gi, name)]
    H = _H_of_last_iteration(al, backend, d, GOLD["a%d_guess" % gi])
    assert np.max(np.abs(H - Hg)) / np.max(np.abs(Hg)) < 1e-5  # float32 J entries vs float64 J (both backends)
    assert np.max(np.abs(al.moving_in_fixed() - X_gold)) <= 1e-5  # (the batch of one ends on the same estimate)
    if backend == "oracle":
        H, b, dx = al.last_system()
        bg, dxg = GOLD["a%d_%s_b" % (gi, name)], GOLD["a%d_%s_dx" % (gi, name)]
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
    H = _H_of_last_iteration(al, backend, d, syn.identity(2))
    assert np.max(np.abs(H - GOLD["c_H"])) / np.max(np.abs(GOLD["c_H"])) < 1e-5
    if backend == "oracle":
        H, b, dx = al.last_system()
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
    H = _H_of_last_iteration(al, backend, d, GOLD_FD["r_guess"])
    assert np.max(np.abs(H - GOLD_FD["r_H"])) / np.max(np.abs(GOLD_FD["r_H"])) < 1e-4
    if backend == "oracle":
        H, b, dx = al.last_system()
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
    H = _H_of_last_iteration(al, backend, d, GOLD_FD["e%d_guess" % gi])
    assert np.max(np.abs(H - GOLD_FD["e%d_H" % gi])) / np.max(np.abs(GOLD_FD["e%d_H" % gi])) < 1e-5
    if backend == "oracle":
        H, b, dx = al.last_system()
        assert np.max(np.abs(dx - GOLD_FD["e%d_dx" % gi])) < 1e-5


# ---- whole compute() calls (tests/golden/make_golden2.py): robustifier variants, inlier-only run, SE(2) plane, prior + cue
GOLD2 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "icp_golden2.npz"))


def _check_stats(stats, gold_rows, chi_rtol=2e-4):
    """IterationStats of every iteration against the golden's [num_inliers, num_outliers, num_correspondences,
    chi_inliers, chi_outliers]: counts exactly, chi sums to float32 accumulation accuracy"""
    assert len(stats) == len(gold_rows)
    for it, (s, gr) in enumerate(zip(stats, gold_rows)):
        assert s["iteration"] == it and s["solver_status"] == 0
        assert (s["num_inliers"], s["num_outliers"], s["num_correspondences"]) == (int(gr[0]), int(gr[1]), int(gr[2])), (it, s, gr)
        assert abs(s["chi_inliers"] - gr[3]) <= chi_rtol * max(gr[3], 1e-6), (it, s, gr)
        assert abs(s["chi_outliers"] - gr[4]) <= chi_rtol * max(gr[4], 1e-6), (it, s, gr)


@pytest.mark.parametrize("backend", BACKENDS)
def test_saturated_robustifier_then_inlier_only_clamp_run(backend, oracle):
    """Saturated kernel for two iterations, then the inlier-only run with Clamp robustifiers
    (multi_aligner_impl.cpp:163-211): statistics of all four iterations, the final H, estimate and correspondences"""
    d = syn.cloud_pair_3d(n=2500, seed=777)
    al = _aligner(backend, oracle, abi.SE3_QUAT_RIGHT)
    al.set_params(max_iterations=2, min_num_inliers=10, enable_inlier_only_runs=True)
    setup_pair(al, d, cue_config(abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE, 0.25, abi.ROBUST_SATURATED, 0.0008))
    assert al.compute() == abi.SUCCESS
    _check_stats(al.iteration_stats(), GOLD2["s_stats"])
    assert np.max(np.abs(al.moving_in_fixed() - GOLD2["s_X"])) <= 1e-5
    # the correspondences of the last iteration: same pairs; the responses differ from the golden's by the float32
    # rounding of the estimate they were searched with (three Gauss-Newton steps in, the estimates agree to ~1e-7)
    c, idx, d2 = al.correspondences(0), GOLD2["s_idx"], GOLD2["s_d2"]
    sel = idx >= 0
    assert np.array_equal(c["moving_idx"], np.nonzero(sel)[0].astype(np.int32))
    assert np.array_equal(c["fixed_idx"], idx[sel])
    assert np.max(np.abs(c["response"] - d2[sel])) <= 1e-6
    H = _H_of_last_iteration(al, backend, d, syn.identity(3))
    assert np.max(np.abs(H - GOLD2["s_H"])) / np.max(np.abs(GOLD2["s_H"])) < 1e-4
    _check_stats(al.iteration_stats(), GOLD2["s_stats"])  # (the batch of one leaves the same statistics behind)


@pytest.mark.parametrize("backend", BACKENDS)
def test_se2_point_to_plane(backend, oracle):
    d = syn.scan_pair_2d(beams=1000, seed=1200)
    al = _aligner(backend, oracle, abi.SE2_RIGHT)
    al.set_params(max_iterations=1)
    setup_pair(al, d, cue_config(abi.SE2_RIGHT, abi.SLICE_P2PLANE, 0.5))
    assert al.compute() == abi.SUCCESS
    _check_corr(al.correspondences(0), GOLD2["l_idx"], GOLD2["l_d2"])
    assert np.max(np.abs(al.moving_in_fixed() - GOLD2["l_X"])) <= 1e-5
    st = al.iteration_stats()
    assert len(st) == 1 and st[0]["num_inliers"] == int(np.sum(GOLD2["l_idx"] >= 0)) and st[0]["num_outliers"] == 0
    assert abs(st[0]["chi_inliers"] - float(GOLD2["l_chi"])) <= 2e-4 * float(GOLD2["l_chi"])
    H = _H_of_last_iteration(al, backend, d, syn.identity(2))
    assert np.max(np.abs(H - GOLD2["l_H"])) / np.max(np.abs(GOLD2["l_H"])) < 1e-5
    if backend == "oracle":
        H, b, dx = al.last_system()
        assert np.max(np.abs(b - GOLD2["l_b"])) / np.max(np.abs(GOLD2["l_b"])) < 1e-4
        assert np.max(np.abs(dx - GOLD2["l_dx"])) < 1e-5


@pytest.mark.parametrize("backend", BACKENDS)
def test_prior_slice_next_to_a_cue_slice(backend, oracle):
    """H and b summed over a point-to-point cue slice and an odometry-prior slice (multi_aligner_impl.cpp:144-160); the
    golden linearises the prior e = t2v(Z^-1 X) by finite differences"""
    from helpers import prior_config

    d = syn.cloud_pair_3d(n=1500, seed=888)
    al = _aligner(backend, oracle, abi.SE3_QUAT_RIGHT)
    al.set_params(max_iterations=1, min_num_inliers=10)
    setup_pair(al, d, cue_config(abi.SE3_QUAT_RIGHT, abi.SLICE_P2P, 0.25))
    pi = al.add_slice(prior_config(abi.SE3_QUAT_RIGHT, info=[float(v) for v in GOLD2["p_info"]], sets_guess=0))
    al.set_prior_measurement(pi, GOLD2["p_Z"])
    al.set_moving_in_fixed(syn.identity(3))
    assert al.compute() == abi.SUCCESS
    _check_corr(al.correspondences(0), GOLD2["p_idx"], GOLD2["p_d2"])
    assert np.max(np.abs(al.moving_in_fixed() - GOLD2["p_X"])) <= 1e-5
    st = al.iteration_stats()
    assert len(st) == 1 and st[0]["num_inliers"] == int(GOLD2["p_num_inliers"]) and st[0]["num_outliers"] == 0
    assert st[0]["num_correspondences"] == int(np.sum(GOLD2["p_idx"] >= 0)) + 1  # priors count one (:275-285)
    assert abs(st[0]["chi_inliers"] - float(GOLD2["p_chi_inliers"])) <= 2e-4 * float(GOLD2["p_chi_inliers"])
    H = _H_of_last_iteration(al, backend, d, syn.identity(3))
    assert np.max(np.abs(H - GOLD2["p_H"])) / np.max(np.abs(GOLD2["p_H"])) < 1e-5
    if backend == "oracle":
        H, b, dx = al.last_system()
        assert np.max(np.abs(b - GOLD2["p_b"])) / np.max(np.abs(GOLD2["p_b"])) < 1e-4
        assert np.max(np.abs(dx - GOLD2["p_dx"])) < 1e-5


# ---- VERDICT r3 #6 (tests/golden/make_golden3.py): C3's two slices summed at 160 x 120; the termination criterion firing
GOLD3 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "icp_golden3.npz"))


@pytest.mark.parametrize("backend", BACKENDS)
def test_c3_two_projective_slices_summed(backend, oracle):
    """MultiAligner with a projective point-to-plane slice and a projective reprojection slice (BASELINE config C3, here at
    160 x 120): H and b of the two factors are ADDED into one system (multi_aligner_impl.cpp:144-160).  Golden: restated
    finder, matrix-form point-to-plane Jacobian, finite-difference reprojection Jacobian, one Gauss-Newton step."""
    from helpers import projective_config

    d = syn.rgbd_pair(rows=120, cols=160, seed=3200)
    al = _aligner(backend, oracle, abi.SE3_QUAT_RIGHT)
    al.set_params(max_iterations=1, min_num_inliers=10)
    for sk in (abi.SLICE_P2PLANE, abi.SLICE_REPROJECTION):
        si = al.add_slice(projective_config(abi.SE3_QUAT_RIGHT, sk, d, gate=0.05))
        al.set_fixed(si, d["fixed"], d["fixed_normals"])
        al.set_moving(si, d["moving"], d["moving_normals"])
    al.set_moving_in_fixed(GOLD3["t_guess"])
    assert al.compute() == abi.SUCCESS
    match = GOLD3["t_match"]
    sel = match >= 0
    for si in (0, 1):  # both slices run the same finder on the same clouds
        c = al.correspondences(si)
        assert np.array_equal(c["moving_idx"], np.nonzero(sel)[0].astype(np.int32))
        assert np.array_equal(c["fixed_idx"], match[sel])
        assert c["response"].tobytes() == GOLD3["t_resp"][sel].tobytes()
    st = al.iteration_stats()
    assert len(st) == 1 and st[0]["num_inliers"] == int(GOLD3["t_n1"]) + int(GOLD3["t_n2"]) and st[0]["num_outliers"] == 0
    assert st[0]["num_correspondences"] == 2 * int(sel.sum())
    assert abs(st[0]["chi_inliers"] - float(GOLD3["t_chi"])) <= 1e-3 * float(GOLD3["t_chi"])
    assert np.max(np.abs(al.moving_in_fixed() - GOLD3["t_X"])) <= 1e-5
    H = al.last_system()[0] if backend == "oracle" else al.information().astype(np.float64)
    Hg = GOLD3["t_H"]
    assert np.max(np.abs(H - Hg)) / np.max(np.abs(Hg)) < 1e-4
    # (the sum is a sum: either slice alone lands somewhere else -- the reprojection slice's H is 10^4 times the plane
    # slice's in size, pixel units against metres, so the estimate, not H, is where the plane slice's share shows)
    assert np.max(np.abs(GOLD3["t_X"] - GOLD3["t_X_plane_only"])) > 1e-4
    assert np.max(np.abs(GOLD3["t_X"] - GOLD3["t_X_reprojection_only"])) > 1e-4
    if backend == "oracle":
        H, b, dx = al.last_system()
        assert np.max(np.abs(b - GOLD3["t_b"])) / np.max(np.abs(GOLD3["t_b"])) < 1e-3
        assert np.max(np.abs(dx - GOLD3["t_dx"])) < 1e-5


@pytest.mark.parametrize("backend", BACKENDS)
def test_termination_criterion_fires_with_its_quirks(backend, oracle):
    """AlignerTerminationCriteriaStandard_ (aligner_termination_criteria_impl.cpp:24-65), window 5, on a 12-iteration
    compute(): the golden restates the criterion WITH its quirks -- line 46 holds the outlier window against
    num_correspondences_range, line 53 the chi window against num_outliers_range -- and picks parameters under which a
    criterion without them would stop one iteration later.  Iteration count, every IterationStats, the estimate."""
    d = syn.cloud_pair_3d(n=3000, seed=4400, noise_sigma=0.004)
    al = _aligner(backend, oracle, abi.SE3_QUAT_RIGHT)
    al.set_params(max_iterations=12, min_num_inliers=10)
    t = abi.default_termination_params()
    t.window_size, t.num_correspondences_range, t.num_inliers_range, t.num_outliers_range = [int(v) for v in GOLD3["c_params"]]
    t.chi_epsilon = float(GOLD3["c_eps"])
    al.set_termination_criteria(t)
    setup_pair(al, d, cue_config(abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE, 0.25, abi.ROBUST_CAUCHY, 0.0004), with_moving_normals=False)
    assert al.compute() == abi.SUCCESS
    st = al.iteration_stats()
    stop = int(GOLD3["c_stop"])
    assert int(GOLD3["c_stop_without_quirks"]) != stop
    assert len(st) == stop + 1, (len(st), stop)  # (the loop breaks after the iteration at which hasToStop() says so)
    _check_stats(st, GOLD3["c_stats"])
    assert np.max(np.abs(al.moving_in_fixed() - GOLD3["c_X"])) <= 1e-5
