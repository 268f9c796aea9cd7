#!/usr/bin/env python
"""Generate tests/golden/icp_golden_fd.npz: a SECOND, derivative-free derivation of the two pieces of factor arithmetic
that tests/golden/make_golden.py does not reach and that only this build's specification pins (the reference's factors
live in srrg2_solver, which is not under /root/reference):

  * the pinhole REPROJECTION factor behind the PROJECTIVE finder (BASELINE config C3): the finder is restated from
    DESIGN.md section 5 (float32 projection, z-buffer = minimum depth then minimum index, depth / distance gates), the
    Jacobian of e(dx) = pi(X v2t(dx) p) - pi(f) is taken by CENTRAL FINITE DIFFERENCES of the residual function in
    float64 -- no analytic derivative of the projection is written on the golden side;
  * the SE(3) EULER (+) of MultiAligner3D (VariableSE3EulerRightAD): X <- X * [Rx(a) Ry(b) Rz(c) | t], composed from
    three elementary rotation matrices, with a finite-difference Jacobian of the point-to-point residual.

Inputs are regenerated from seeds by srrg2_slam_interfaces_amd.synthetic.  Run from the repo root:
  python tests/golden/make_golden_fd.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from srrg2_slam_interfaces_amd import synthetic as syn  # noqa: E402

F = np.float32
EPS = 1e-6


def transform_f32(X, P):
    X = X.astype(F)
    return np.stack([((X[i, 0] * P[:, 0] + X[i, 1] * P[:, 1]) + X[i, 2] * P[:, 2]) + X[i, 3] for i in range(3)], 1)


def quat_v2t(v):
    x, y, z = v[3:]
    w = np.sqrt(1 - x * x - y * y - z * z)
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    T = np.zeros((3, 4))
    T[:, :3], T[:, 3] = R, v[:3]
    return T


def euler_v2t(v):
    a, b, c = v[3:]
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    Rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    T = np.zeros((3, 4))
    T[:, :3], T[:, 3] = Rx @ Ry @ Rz, v[:3]
    return T


def fd_jacobian(residual, n_rows):
    J = np.zeros((n_rows, 6))
    for a in range(6):
        d = np.zeros(6)
        d[a] = EPS
        J[:, a] = (residual(d) - residual(-d)) / (2 * EPS)
    return J


def projective_finder(X, d, gate):
    """float32, as DESIGN.md section 5: pixel = (floor(v + 0.5), floor(u + 0.5)) of u = (K0 qx) / qz + K2; per pixel the
    moving point of minimum depth, ties to the smaller index; kept iff the fixed pixel is valid, |f_z - q_z| <= gate and
    |q - f| <= 2 gate.  Returns fixed pixel per moving point (-1: none) and the response |f_z - q_z|."""
    K, rows, cols = d["K"].astype(F), d["rows"], d["cols"]
    Q = transform_f32(X, d["moving"])
    qz = Q[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        u = (K[0, 0] * Q[:, 0]) / qz + K[0, 2]
        v = (K[1, 1] * Q[:, 1]) / qz + K[1, 2]
    uf, vf = u + F(0.5), v + F(0.5)
    ok = np.isfinite(Q).all(1) & (qz >= F(d["depth_min"])) & (qz <= F(d["depth_max"]))
    ok &= (uf >= 0) & (uf < F(cols)) & (vf >= 0) & (vf < F(rows))
    pix = np.where(ok, np.floor(np.where(ok, vf, 0)).astype(np.int64) * cols + np.floor(np.where(ok, uf, 0)).astype(np.int64), -1)
    # z-buffer: minimum (depth, index) per pixel
    order = np.lexsort((np.arange(len(pix)), qz, pix))
    order = order[pix[order] >= 0]
    first = np.ones(len(order), bool)
    first[1:] = pix[order][1:] != pix[order][:-1]
    winners = order[first]
    match = np.full(len(pix), -1, np.int64)
    resp = np.zeros(len(pix), F)
    f = d["fixed"][pix[winners]]
    q = Q[winners]
    valid = np.isfinite(f).all(1)
    dd = np.abs(f[:, 2] - q[:, 2])
    dx, dy, dz = f[:, 0] - q[:, 0], f[:, 1] - q[:, 1], f[:, 2] - q[:, 2]
    d2 = (dx * dx + dy * dy) + dz * dz
    g2 = (F(2) * F(gate)) * (F(2) * F(gate))
    with np.errstate(invalid="ignore"):
        keep = valid & (dd <= F(gate)) & (d2 <= g2)
    match[winners[keep]] = pix[winners[keep]]
    resp[winners[keep]] = dd[keep]
    return match, resp, Q


def gn_reprojection(X, d, match):
    """one Gauss-Newton step of the reprojection slice, quaternion (+), float64, finite-difference Jacobians"""
    K = d["K"].astype(np.float64)
    X64 = X.astype(np.float64)

    def pi(p):
        return np.array([K[0, 0] * p[0] / p[2] + K[0, 2], K[1, 1] * p[1] / p[2] + K[1, 2]])

    H, b, n = np.zeros((6, 6)), np.zeros(6), 0
    for i in np.nonzero(match >= 0)[0]:
        p = d["moving"][i].astype(np.float64)
        f = d["fixed"][match[i]].astype(np.float64)
        if not f[2] > 0:
            continue

        def residual(dx):
            T = syn.se3_mul(X64, quat_v2t(dx))
            return pi(T[:, :3] @ p + T[:, 3]) - pi(f)

        e = residual(np.zeros(6))
        if np.max(np.abs(e)) > 8.0:
            continue
        J = fd_jacobian(residual, 2)
        H += J.T @ J
        b += J.T @ e
        n += 1
    dx = np.linalg.solve(H, -b)
    return H, b, dx, syn.se3_mul(X64, quat_v2t(dx)).astype(F), n


def nn_bruteforce(Q, Fx, gate):
    idx = np.full(Q.shape[0], -1, np.int32)
    d2o = np.zeros(Q.shape[0], F)
    gate2 = F(gate) * F(gate)
    for i in range(Q.shape[0]):
        dd = Fx - Q[i]
        d2 = (dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1]) + dd[:, 2] * dd[:, 2]
        j = int(np.argmin(d2))
        if d2[j] <= gate2:
            idx[i], d2o[i] = j, d2[j]
    return idx, d2o


def gn_euler_p2p(X, d, idx):
    X64 = X.astype(np.float64)
    H, b = np.zeros((6, 6)), np.zeros(6)
    for i in np.nonzero(idx >= 0)[0]:
        p = d["moving"][i].astype(np.float64)
        f = d["fixed"][idx[i]].astype(np.float64)

        def residual(dx):
            T = syn.se3_mul(X64, euler_v2t(dx))
            return T[:, :3] @ p + T[:, 3] - f

        e = residual(np.zeros(6))
        J = fd_jacobian(residual, 3)
        H += J.T @ J
        b += J.T @ e
    dx = np.linalg.solve(H, -b)
    return H, b, dx, syn.se3_mul(X64, euler_v2t(dx)).astype(F)


def main():
    out = {}
    d = syn.rgbd_pair(rows=60, cols=80, seed=3100)
    guess = syn.se3(np.array([0.01, 0.0, -0.01]), np.deg2rad([0.2, 0.4, -0.1])).astype(F)
    match, resp, _ = projective_finder(guess, d, 0.05)
    H, b, dx, Xn, n = gn_reprojection(guess, d, match)
    out.update(r_guess=guess, r_match=match.astype(np.int32), r_resp=resp, r_H=H, r_b=b, r_dx=dx, r_X=Xn, r_n=n)
    d3 = syn.cloud_pair_3d(n=1500, seed=321)
    for gi, X in enumerate([syn.identity(3), syn.se3(np.array([0.02, 0.03, -0.01]), np.deg2rad([1.0, -0.7, 0.4])).astype(F)]):
        idx, d2 = nn_bruteforce(transform_f32(X, d3["moving"]), d3["fixed"], 0.25)
        H, b, dx, Xn = gn_euler_p2p(X, d3, idx)
        out.update({"e%d_guess" % gi: X, "e%d_idx" % gi: idx, "e%d_d2" % gi: d2, "e%d_H" % gi: H, "e%d_b" % gi: b,
                    "e%d_dx" % gi: dx, "e%d_X" % gi: Xn})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "icp_golden_fd.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; reprojection factors:", n, "matches:", int((match >= 0).sum()))


if __name__ == "__main__":
    main()
