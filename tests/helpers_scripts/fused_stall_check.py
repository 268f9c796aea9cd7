"""Run by tests/test_gpu_fused_control.py::test_polling_waves_apply_the_control_step_themselves in a process whose product
library is the -DSRRG2_FUSED_STALL build (SRRG2_AMD_LIB): the designated wave of every fused control step sleeps ~0.3 ms before
it computes, so every polling wave runs into SRRG2_FUSED_POLL_LIMIT and applies the step itself, in registers
(wave_control<.., false>, csrc/kernels.hip).  Every scenario must give the oracle's bits, and the fallback counter must move."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
try:
    import torch  # noqa: F401  (before the product library: tests/conftest.py)
except Exception:
    pass

import srrg2_slam_interfaces_amd as product  # noqa: E402
from helpers import assert_same_run, cue_config, prior_config, setup_pair  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402
from srrg2_slam_interfaces_amd import _abi as abi  # noqa: E402
from srrg2_slam_interfaces_amd import _capi  # noqa: E402
from srrg2_slam_interfaces_amd import synthetic as syn  # noqa: E402

lib = _capi.lib()
assert "stall" in _capi.LIB_PATH, _capi.LIB_PATH
FUSED = {"search_lists": 2, "fused_control": 1}


def fallbacks(reset=False):
    n = C.c_ulonglong(0)
    assert lib.srrg2_amd_debug_fused_fallbacks(C.byref(n), 1 if reset else 0) == 0
    return n.value


def single(kind, slice_kind):
    if kind == abi.SE2_RIGHT:
        d = syn.scan_pair_2d(beams=3000, sigma=0.01, seed=1234)
        gate, thr = 0.5, 0.002
    else:
        d = syn.cloud_pair_3d(n=15000, seed=2200, noise_sigma=0.01)
        gate, thr = 0.25, 0.0005
    cfg = cue_config(kind, slice_kind, gate, abi.ROBUST_CAUCHY, thr)
    runs = []
    for al in (oracle.OracleAligner(kind), product.MultiAligner(kind)):
        if not isinstance(al, oracle.OracleAligner):
            al.set_tuning(**FUSED)
        al.set_params(max_iterations=12, min_num_inliers=10, enable_inlier_only_runs=True, keep_only_inlier_correspondences=True)
        al.set_termination_criteria(abi.default_termination_params())
        setup_pair(al, d, cfg)
        al.compute()
        al.set_moving_in_fixed(syn.identity(al.dim))
        al.compute()
        runs.append(al)
    assert runs[0].status() == abi.SUCCESS
    assert_same_run(runs[0], runs[1])


def batch():
    kind = abi.SE3_QUAT_RIGHT
    probs = syn.batch_3d(K=12, n=9000, seed=8800, shared_fixed_group=64, t_max=0.1, rpy_max_deg=2.0)
    fixed, fixed_n = probs[0]["fixed"], probs[0]["fixed_normals"]
    movs = [p["moving"] for p in probs]
    nrms = [p["moving_normals"] for p in probs]
    movs[1], nrms[1] = movs[1][:0], nrms[1][:0]  # (an empty cloud: Fail without statistics)
    guesses = [syn.identity(3)] * len(probs)
    cfg = cue_config(kind, abi.SLICE_P2PLANE, 0.3, abi.ROBUST_CAUCHY, 0.05, 0.7)
    out = []
    for al in (oracle.OracleAligner(kind), product.MultiAligner(kind)):
        if not isinstance(al, oracle.OracleAligner):
            al.set_tuning(**FUSED)
        al.set_params(max_iterations=7, min_num_inliers=100)
        al.add_slice(cfg)
        al.set_fixed(0, fixed, fixed_n)
        out.append(al.compute_batch(movs, guesses, nrms))
    for r, g in zip(*out):
        assert r["status"] == g["status"] and r["num_iterations"] == g["num_iterations"]
        assert r["moving_in_fixed"].tobytes() == g["moving_in_fixed"].tobytes()
        assert r["last"] == g["last"] and r["num_correspondences"] == g["num_correspondences"]
        assert np.asarray(r["information"]).tobytes() == np.asarray(g["information"]).tobytes()


def projective():
    kind = abi.SE3_QUAT_RIGHT
    r = syn.rgbd_pair(rows=120, cols=160, seed=3100)
    runs = []
    for al in (oracle.OracleAligner(kind), product.MultiAligner(kind)):
        if not isinstance(al, oracle.OracleAligner):
            al.set_tuning(fused_control=1)
        al.set_params(max_iterations=8, min_num_inliers=10, enable_inlier_only_runs=True)
        for sk in (abi.SLICE_P2PLANE, abi.SLICE_REPROJECTION):
            c = abi.default_slice_config(kind)
            c.kind, c.finder, c.finder_max_distance = sk, abi.FINDER_PROJECTIVE, 0.05
            c.robustifier, c.robustifier_chi_threshold = abi.ROBUST_CAUCHY, 0.05 if sk == abi.SLICE_P2PLANE else 4.0
            for i, v in enumerate(r["K"].reshape(-1)):
                c.camera_matrix[i] = v
            c.image_rows, c.image_cols, c.depth_min, c.depth_max = r["rows"], r["cols"], r["depth_min"], r["depth_max"]
            si = al.add_slice(c)
            if si == 0 or isinstance(al, oracle.OracleAligner):
                al.set_fixed(si, r["fixed"], r["fixed_normals"])
                al.set_moving(si, r["moving"], r["moving_normals"])
            else:
                al.share_clouds(si, 0)
        al.set_moving_in_fixed(syn.identity(3))
        al.compute()
        runs.append(al)
    assert runs[0].status() == abi.SUCCESS
    assert_same_run(runs[0], runs[1], slices=(0, 1))


def prior_cue(kind, slice_kind):
    """a prior slice before and one behind the cue slice: the polling waves linearise the prior factors too (wave_prior)"""
    if kind == abi.SE2_RIGHT:
        d = syn.scan_pair_2d(beams=3000, sigma=0.01, seed=1234)
        gate, thr = 0.5, 0.002
        Za, Zb = syn.se2(0.03, -0.02, 0.01).astype(np.float32), syn.se2(-0.05, 0.04, -0.02).astype(np.float32)
        info_a, info_b = [10.0, 10.0, 100.0], [3.0, 5.0, 40.0]
    else:
        d = syn.cloud_pair_3d(n=15000, seed=2200, noise_sigma=0.01)
        gate, thr = 0.25, 0.0005
        Za = syn.se3(np.array([0.04, -0.02, 0.01]), np.deg2rad([0.5, -1.0, 1.5])).astype(np.float32)
        Zb = syn.se3(np.array([-0.03, 0.05, 0.02]), np.deg2rad([-1.0, 0.5, -0.7])).astype(np.float32)
        info_a, info_b = [10, 10, 10, 100, 100, 100], [3, 4, 5, 30, 40, 50]
    cfg = cue_config(kind, slice_kind, gate, abi.ROBUST_CAUCHY, thr)
    runs = []
    for al in (oracle.OracleAligner(kind), product.MultiAligner(kind)):
        if not isinstance(al, oracle.OracleAligner):
            al.set_tuning(**FUSED)
        al.set_params(max_iterations=12, min_num_inliers=10, enable_inlier_only_runs=True)
        al.set_termination_criteria(abi.default_termination_params())
        pa = al.add_slice(prior_config(kind, info=info_a, sets_guess=1))
        cue = setup_pair(al, d, cfg)
        pb = al.add_slice(prior_config(kind, info=info_b, sets_guess=0))
        al.set_prior_measurement(pa, Za)
        al.set_prior_measurement(pb, Zb)
        al.compute()
        al.set_moving_in_fixed(syn.identity(al.dim))
        al.compute()
        runs.append(al)
    assert runs[0].status() == abi.SUCCESS
    assert_same_run(runs[0], runs[1], slices=(cue,))


fallbacks(reset=True)
counts = {}
for name, fn in (("se3_plane", lambda: single(abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE)), ("se2_p2p", lambda: single(abi.SE2_RIGHT, abi.SLICE_P2P)),
                 ("batch", batch), ("projective", projective),
                 ("prior_cue_se3", lambda: prior_cue(abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE)),
                 ("prior_cue_se2", lambda: prior_cue(abi.SE2_RIGHT, abi.SLICE_P2P))):
    fn()
    counts[name] = fallbacks(reset=True)
print("fused-control fallbacks per scenario:", counts)
assert all(v > 0 for v in counts.values()), counts
print("STALL-OK")
