#!/usr/bin/env python
"""Generate tests/golden/icp_golden2.npz: more independent anchors of the aligner path (VERDICT r2 #5b).

Like make_golden.py this is a SECOND, independently written numpy restatement -- brute-force float32 nearest neighbour,
float64 Gauss-Newton steps in matrix form -- but of whole compute() calls:
  s_*  Saturated robustifier + inlier-only second run with Clamp (multi_aligner_impl.cpp:163-211), SE(3) point-to-plane,
       2 + 2 iterations: IterationStats of every iteration, the last H, the final estimate and correspondences;
  l_*  SE(2) point-to-plane (a laser scan against wall normals), one iteration;
  p_*  a prior slice next to a cue slice (AlignerSliceOdom3DPrior + point-to-point cue: H and b are summed over the
       slices, multi_aligner_impl.cpp:144-160), one iteration; the prior factor e = t2v(Z^-1 X) is linearised by
       CENTRAL FINITE DIFFERENCES of the error function (no analytic Jacobian on the golden side).
Inputs are regenerated from seeds (srrg2_slam_interfaces_amd.synthetic), not stored.

Run from the repo root:  python tests/golden/make_golden2.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import nn_bruteforce, quat_v2t, skew, transform_f32  # noqa: E402
from srrg2_slam_interfaces_amd import synthetic as syn  # noqa: E402

F = np.float32


def R_to_quat(R):
    """unit quaternion (w, x, y, z) of a rotation matrix, w >= 0 (largest-pivot form)"""
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    cands = [tr, R[0, 0] - R[1, 1] - R[2, 2], R[1, 1] - R[0, 0] - R[2, 2], R[2, 2] - R[0, 0] - R[1, 1]]
    k = int(np.argmax(cands))
    s = 2.0 * np.sqrt(1.0 + cands[k])
    if k == 0:
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    elif k == 1:
        q = np.array([(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s])
    elif k == 2:
        q = np.array([(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s])
    else:
        q = np.array([(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s])
    q = q / np.linalg.norm(q)
    return -q if q[0] < 0 else q


def quat_to_R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def fix_transform(X):
    """fixTransform (multi_aligner_impl.cpp:91-93): the rotation re-normalised through a unit quaternion"""
    Y = X.astype(np.float64).copy()
    Y[:, :3] = quat_to_R(R_to_quat(Y[:, :3]))
    return Y.astype(F)


def t2v(T):
    return np.concatenate([T[:, 3], R_to_quat(T[:, :3])[1:]])


def robust_weight(kind, thr, chi):
    """(weight, kernelized): inlier iff chi < thr"""
    if kind == "none" or chi < thr:
        return 1.0, False
    if kind == "clamp":
        return 0.0, True
    if kind == "saturated":
        return float(F(thr) / F(chi)), True
    return float(F(1.0) / (F(1.0) + F(chi) / F(thr))), True  # cauchy


def cue_system_3d(X, d, idx, plane, kind, thr):
    """H, b and the statistics of one linearisation of an SE(3) cue slice (quaternion-right perturbation, k = 2)"""
    X64 = X.astype(np.float64)
    R = X64[:, :3]
    H, b = np.zeros((6, 6)), np.zeros(6)
    n_in = n_out = 0
    chi_in = chi_out = 0.0
    Q = transform_f32(X, d["moving"])
    for i in np.nonzero(idx >= 0)[0]:
        p = d["moving"][i].astype(np.float64)
        f = d["fixed"][idx[i]].astype(np.float64)
        q = Q[i].astype(np.float64)
        Jp = np.hstack([R, -2.0 * R @ skew(p)])
        if plane:
            n = d["fixed_normals"][idx[i]].astype(np.float64)
            e = np.array([F(n @ (q - f))], np.float64)
            J = (n @ Jp)[None, :]
        else:
            e, J = q - f, Jp
        chi = float(F(e.astype(F) @ e.astype(F)))
        w, kern = robust_weight(kind, thr, chi)
        if kern:
            n_out += 1
            chi_out += chi
        else:
            n_in += 1
            chi_in += chi
        H += w * J.T @ J
        b += w * J.T @ e
    return H, b, n_in, n_out, chi_in, chi_out


def run_saturated_then_clamp(d, gate, thr, iterations):
    """compute() with a Saturated robustifier and enable_inlier_only_runs: `iterations` Gauss-Newton iterations with
    the configured kernel, then `iterations` more with Clamp (multi_aligner_impl.cpp:163-181)"""
    X = syn.identity(3).astype(F)
    stats = []
    H = idx = d2 = None
    for kind in ("saturated", "clamp"):
        for _ in range(iterations):
            idx, d2 = nn_bruteforce(transform_f32(X, d["moving"]), d["fixed"], gate)
            H, b, n_in, n_out, chi_in, chi_out = cue_system_3d(X, d, idx, True, kind, thr)
            dx = np.linalg.solve(H, -b)
            X = syn.se3_mul(X.astype(np.float64), quat_v2t(dx)).astype(F)
            stats.append([n_in, n_out, int(np.sum(idx >= 0)), chi_in, chi_out])
    return fix_transform(X), np.array(stats, np.float64), H, idx, d2


def se2_plane_step(d, gate):
    X = syn.identity(2).astype(F)
    idx, d2 = nn_bruteforce(transform_f32(X, d["moving"]), d["fixed"], gate)
    X64 = X.astype(np.float64)
    R = X64[:2, :2]
    H, b = np.zeros((3, 3)), np.zeros(3)
    Q = transform_f32(X, d["moving"])
    chi_sum = 0.0
    for i in np.nonzero(idx >= 0)[0]:
        p = d["moving"][i].astype(np.float64)
        f = d["fixed"][idx[i]].astype(np.float64)
        n = d["fixed_normals"][idx[i]].astype(np.float64)
        e = float(F(n @ (Q[i].astype(np.float64) - f)))
        Jp = np.hstack([R, (R @ np.array([-p[1], p[0]]))[:, None]])  # d(X exp(dx) p)/d dx, dx = (tx, ty, theta)
        J = n @ Jp
        H += np.outer(J, J)
        b += J * e
        chi_sum += float(F(e) * F(e))
    dx = np.linalg.solve(H, -b)
    Xn = (X64 @ syn.se2(dx[0], dx[1], dx[2])).astype(F)
    return idx, d2, H, b, dx, Xn, chi_sum


def prior_system_fd(X, Z, info):
    """H, b, chi of e(dx) = t2v(Z^-1 X v2t(dx)) at dx = 0 with a central-difference Jacobian"""
    X64, Z64 = X.astype(np.float64), Z.astype(np.float64)
    Zinv = syn.se3_inv(Z64)

    def err(dx):
        return t2v(syn.se3_mul(Zinv, syn.se3_mul(X64, quat_v2t(dx))))

    e = err(np.zeros(6))
    J = np.zeros((6, 6))
    h = 1e-6
    for k in range(6):
        dp, dm = np.zeros(6), np.zeros(6)
        dp[k], dm[k] = h, -h
        J[:, k] = (err(dp) - err(dm)) / (2 * h)
    Om = np.diag(np.asarray(info, np.float64))
    return J.T @ Om @ J, J.T @ Om @ e, float(e @ Om @ e)


def main():
    out = {}
    # ---- Saturated + inlier-only Clamp run, 2 + 2 iterations
    d3 = syn.cloud_pair_3d(n=2500, seed=777)
    X, stats, H, idx, d2 = run_saturated_then_clamp(d3, 0.25, 0.0008, 2)
    out.update(s_X=X, s_stats=stats, s_H=H, s_idx=idx, s_d2=d2)
    # ---- SE(2) point-to-plane
    d2d = syn.scan_pair_2d(beams=1000, seed=1200)
    idx, r2, H, b, dx, Xn, chi = se2_plane_step(d2d, 0.5)
    out.update(l_idx=idx, l_d2=r2, l_H=H, l_b=b, l_dx=dx, l_X=Xn, l_chi=chi)
    # ---- prior slice + cue slice
    dp = syn.cloud_pair_3d(n=1500, seed=888)
    Xg = syn.identity(3).astype(F)
    Z = syn.se3(np.array([0.04, -0.02, 0.03]), np.deg2rad([0.8, -1.2, 1.6])).astype(F)
    info = [400.0, 400.0, 400.0, 900.0, 900.0, 900.0]
    idx, r2 = nn_bruteforce(transform_f32(Xg, dp["moving"]), dp["fixed"], 0.25)
    Hc, bc, n_in, n_out, chi_in, chi_out = cue_system_3d(Xg, dp, idx, False, "none", 1.0)
    Hp, bp, chi_p = prior_system_fd(Xg, Z, info)
    H, b = Hc + Hp, bc + bp
    dx = np.linalg.solve(H, -b)
    Xn = fix_transform(syn.se3_mul(Xg.astype(np.float64), quat_v2t(dx)).astype(F))
    out.update(p_idx=idx, p_d2=r2, p_H=H, p_b=b, p_dx=dx, p_X=Xn, p_Z=Z, p_info=np.array(info, F),
               p_num_inliers=n_in + 1, p_chi_inliers=chi_in + chi_p)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "icp_golden2.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    print("saturated/clamp stats [n_in, n_out, n_corr, chi_in, chi_out]:\n", stats)


if __name__ == "__main__":
    main()
