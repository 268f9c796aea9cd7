"""Shared builders for the parity tests: the same calls drive the oracle and the HIP library."""
import numpy as np

from srrg2_slam_interfaces_amd import _abi as abi
from srrg2_slam_interfaces_amd import synthetic as syn


def cue_config(kind, slice_kind, gate, robust=abi.ROBUST_NONE, thr=0.05, normal_cos=-2.0, min_corr=0):
    c = abi.default_slice_config(kind)
    c.kind = slice_kind
    c.finder = abi.FINDER_NN_GATED
    c.finder_max_distance = gate
    c.robustifier = robust
    c.robustifier_chi_threshold = thr
    c.finder_normal_cos = normal_cos
    c.min_num_correspondences = min_corr
    return c


def prior_config(kind, info=None, sets_guess=1):
    c = abi.default_slice_config(kind)
    c.kind = abi.SLICE_PRIOR
    c.finder = abi.FINDER_NONE
    c.prior_sets_initial_guess = sets_guess
    if info is not None:
        for i, v in enumerate(info):
            c.prior_information_diag[i] = v
    return c


def setup_pair(al, data, cfg, guess=None, with_moving_normals=True):
    si = al.add_slice(cfg)
    al.set_fixed(si, data["fixed"], data.get("fixed_normals"))
    al.set_moving(si, data["moving"], data.get("moving_normals") if with_moving_normals else None)
    al.set_moving_in_fixed(syn.identity(al.dim) if guess is None else guess)
    return si


def assert_same_run(a_ref, a_gpu, slices=(0,), x_tol=1e-5):
    """Compare a finished compute() of the oracle (a_ref) and of the HIP library (a_gpu)."""
    assert a_ref.status() == a_gpu.status()
    s_ref, s_gpu = a_ref.iteration_stats(), a_gpu.iteration_stats()
    assert len(s_ref) == len(s_gpu)
    for r, g in zip(s_ref, s_gpu):
        for key in ("iteration", "num_inliers", "num_outliers", "num_suppressed", "num_correspondences",
                    "solver_status"):
            assert r[key] == g[key], (key, r, g)
        # chi sums are exact fixed-point sums: bit-identical floats
        assert np.float32(r["chi_inliers"]).tobytes() == np.float32(g["chi_inliers"]).tobytes(), (r, g)
        assert np.float32(r["chi_outliers"]).tobytes() == np.float32(g["chi_outliers"]).tobytes(), (r, g)
    X_ref, X_gpu = a_ref.moving_in_fixed(), a_gpu.moving_in_fixed()
    # SE(2)/SE(3) estimate within 1e-5 (BASELINE.json north_star tolerance) ...
    assert np.max(np.abs(X_ref - X_gpu)) <= x_tol, (X_ref, X_gpu)
    # ... and in fact bit-identical, because every sum is an exact fixed-point sum
    assert X_ref.tobytes() == X_gpu.tobytes()
    assert a_ref.num_correspondences() == a_gpu.num_correspondences()
    for si in slices:
        c_ref, c_gpu = a_ref.correspondences(si), a_gpu.correspondences(si)
        assert c_ref.shape == c_gpu.shape
        # indices bit-exact; responses are float32 squared distances computed in the same op order
        assert np.array_equal(c_ref["fixed_idx"], c_gpu["fixed_idx"])
        assert np.array_equal(c_ref["moving_idx"], c_gpu["moving_idx"])
        assert c_ref["response"].tobytes() == c_gpu["response"].tobytes()
        assert np.array_equal(a_ref.factor_status(si), a_gpu.factor_status(si))


def projective_config(kind, slice_kind, data, gate=0.05, robust=abi.ROBUST_NONE, thr=1.0, normal_cos=-2.0):
    c = abi.default_slice_config(kind)
    c.kind = slice_kind
    c.finder = abi.FINDER_PROJECTIVE
    c.finder_max_distance = gate
    c.robustifier = robust
    c.robustifier_chi_threshold = thr
    c.finder_normal_cos = normal_cos
    for i, v in enumerate(np.asarray(data["K"], np.float32).reshape(-1)):
        c.camera_matrix[i] = v
    c.image_rows, c.image_cols = int(data["rows"]), int(data["cols"])
    c.depth_min, c.depth_max = float(data["depth_min"]), float(data["depth_max"])
    return c
