#!/usr/bin/env python
"""Generate tests/golden/icp_golden.npz with an INDEPENDENT numpy implementation of one ICP iteration.

The reference holds no golden vectors for this path ("fixtures / data files / golden
vectors: none"), and it cannot be built or imported here, so the fixtures come from a second, independently
written restatement: brute-force float32 nearest neighbour (numpy element-wise float32 ops = one IEEE
operation each, no FMA, same operation order as DESIGN.md's arithmetic specification) and a float64
Gauss-Newton step with Jacobians written in matrix form (n^T [R | -k R [p]x]) instead of the oracle's
row form.  Inputs are NOT stored: they are regenerated from seeds by srrg2_slam_interfaces_amd.synthetic.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from srrg2_slam_interfaces_amd import synthetic as syn  # noqa: E402

F = np.float32


def transform_f32(X, P):
    """q = ((r0*px + r1*py) + r2*pz) + t, float32, fixed order."""
    X = X.astype(F)
    if P.shape[1] == 3:
        return np.stack([((X[i, 0] * P[:, 0] + X[i, 1] * P[:, 1]) + X[i, 2] * P[:, 2]) + X[i, 3] for i in range(3)], 1)
    return np.stack([(X[i, 0] * P[:, 0] + X[i, 1] * P[:, 1]) + X[i, 2] for i in range(2)], 1)


def nn_bruteforce(Q, Fx, gate):
    """exact gated NN, ties -> smallest fixed index (np.argmin returns the first minimum)."""
    idx = np.full(Q.shape[0], -1, np.int32)
    d2o = np.zeros(Q.shape[0], F)
    gate2 = F(gate) * F(gate)
    for i in range(Q.shape[0]):
        d = Fx - Q[i]
        d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]
        if Q.shape[1] == 3:
            d2 = d2 + d[:, 2] * d[:, 2]
        j = int(np.argmin(d2))
        if d2[j] <= gate2:
            idx[i] = j
            d2o[i] = d2[j]
    return idx, d2o


def skew(p):
    return np.array([[0, -p[2], p[1]], [p[2], 0, -p[0]], [-p[1], p[0], 0]])


def quat_v2t(v):
    x, y, z = v[3:]
    w = np.sqrt(1 - x * x - y * y - z * z)
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    T = np.zeros((3, 4))
    T[:, :3], T[:, 3] = R, v[:3]
    return T


def gn_step_3d(X, d, idx, plane, cauchy_thr=None):
    """one Gauss-Newton step, quaternion-right perturbation (k = 2), float64."""
    X64 = X.astype(np.float64)
    R, t = X64[:, :3], X64[:, 3]
    H = np.zeros((6, 6))
    b = np.zeros(6)
    Q = transform_f32(X, d["moving"])
    for i in np.nonzero(idx >= 0)[0]:
        p = d["moving"][i].astype(np.float64)
        f = d["fixed"][idx[i]].astype(np.float64)
        q = Q[i].astype(np.float64)
        Jp = np.hstack([R, -2.0 * R @ skew(p)])  # d(X exp(dx) p)/d dx
        if plane:
            n = d["fixed_normals"][idx[i]].astype(np.float64)
            e = np.array([F(n @ (q - f))], np.float64)
            J = (n @ Jp)[None, :]
        else:
            e = (q - f)
            J = Jp
        chi = float(F(e.astype(F) @ e.astype(F)))
        w = 1.0
        if cauchy_thr is not None and chi >= cauchy_thr:
            w = float(F(1.0) / (F(1.0) + F(chi) / F(cauchy_thr)))
        H += w * J.T @ J
        b += w * J.T @ e
    dx = np.linalg.solve(H, -b)
    Xn = syn.se3_mul(X64, quat_v2t(dx)).astype(F)
    return H, b, dx, Xn


def gn_step_2d(X, d, idx):
    X64 = X.astype(np.float64)
    R = X64[:2, :2]
    H = np.zeros((3, 3))
    b = np.zeros(3)
    Q = transform_f32(X, d["moving"])
    for i in np.nonzero(idx >= 0)[0]:
        p = d["moving"][i].astype(np.float64)
        f = d["fixed"][idx[i]].astype(np.float64)
        e = Q[i].astype(np.float64) - f
        J = np.hstack([R, (R @ np.array([-p[1], p[0]]))[:, None]])
        H += J.T @ J
        b += J.T @ e
    dx = np.linalg.solve(H, -b)
    Xn = (X64 @ syn.se2(dx[0], dx[1], dx[2])).astype(F)
    return H, b, dx, Xn


def main():
    out = {}
    # case A: C2-shaped, SE(3) point-to-plane, Cauchy; case B: SE(3) point-to-point; at two estimates each
    d3 = syn.cloud_pair_3d(n=2000, seed=123)
    guesses3 = [syn.identity(3), syn.se3(np.array([0.03, -0.02, 0.01]), np.deg2rad([0.5, -1.0, 1.5])).astype(F)]
    for gi, X in enumerate(guesses3):
        idx, d2 = nn_bruteforce(transform_f32(X, d3["moving"]), d3["fixed"], 0.25)
        out["a%d_idx" % gi], out["a%d_d2" % gi] = idx, d2
        for name, plane, thr in (("plane", True, 0.05), ("p2p", False, None)):
            H, b, dx, Xn = gn_step_3d(X, d3, idx, plane, thr)
            out["a%d_%s_H" % (gi, name)], out["a%d_%s_b" % (gi, name)] = H, b
            out["a%d_%s_dx" % (gi, name)], out["a%d_%s_X" % (gi, name)] = dx, Xn
        out["a%d_guess" % gi] = X
    # case C: BASELINE config C1, SE(2) point-to-point, 1k beams
    d2d = syn.scan_pair_2d(beams=1000)
    X = syn.identity(2)
    idx, d2 = nn_bruteforce(transform_f32(X, d2d["moving"]), d2d["fixed"], 0.5)
    H, b, dx, Xn = gn_step_2d(X, d2d, idx)
    out.update(c_idx=idx, c_d2=d2, c_H=H, c_b=b, c_dx=dx, c_X=Xn)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "icp_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
