#!/usr/bin/env python
"""tests/golden/posegraph_golden_c5.npz: ONE Gauss-Newton step of BASELINE's C5 graph at FULL size (50 000 SE(3) poses,
200 000 factors, seed 5000), assembled by an independent, vectorised numpy implementation (central-difference Jacobians of
the residual e = t2v(Z^-1 X_i^-1 X_j), float64) and solved with the sparse DIRECT solver of SciPy (SuperLU, minimum-degree
ordering on A^T + A).  The factorisation takes ~19 minutes and ~4 GB here (400 M non-zeros of fill): run once, the float32
poses after the step are committed (2.4 MB).  Inputs are regenerated from the seed, not stored.
Run from the repo root:  python tests/golden/make_posegraph_golden_c5.py [V E]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from srrg2_slam_interfaces_amd import synthetic as syn  # noqa: E402


def inv(T):
    R = np.swapaxes(T[:, :, :3], 1, 2)
    t = -np.einsum("nij,nj->ni", R, T[:, :, 3])
    return np.concatenate([R, t[:, :, None]], axis=2)


def mul(A, B):
    R = np.einsum("nij,njk->nik", A[:, :, :3], B[:, :, :3])
    t = np.einsum("nij,nj->ni", A[:, :, :3], B[:, :, 3]) + A[:, :, 3]
    return np.concatenate([R, t[:, :, None]], axis=2)


def v2t(v):
    """[t, q.xyz] -> transform; w = sqrt(1 - |q.xyz|^2) (the right perturbation of VariableSE3QuaternionRight)"""
    x, y, z = v[:, 3], v[:, 4], v[:, 5]
    w = np.sqrt(np.maximum(0.0, 1 - x * x - y * y - z * z))
    R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], 1),
                  np.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], 1),
                  np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1)], 1)
    return np.concatenate([R, v[:, :3, None]], axis=2)


def t2v(T):
    R = T[:, :, :3]
    w = np.sqrt(np.maximum(1e-30, 1 + R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2])) / 2
    q = np.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], 1) / (4 * w[:, None])
    return np.concatenate([T[:, :, 3], q], axis=1)


def residual(Xi, Xj, Zinv):
    return t2v(mul(Zinv, mul(inv(Xi), Xj)))


def gauss_newton_step(V, E, seed):
    g = syn.pose_graph_3d(V=V, E=E, seed=seed)
    X = g["poses_init"].astype(np.float64)
    ij = g["ij"]
    V, E = X.shape[0], ij.shape[0]
    Zinv = inv(g["Z"].astype(np.float64))
    Xi, Xj = X[ij[:, 0]], X[ij[:, 1]]
    r0 = residual(Xi, Xj, Zinv)
    chi = float(np.sum(r0 * r0))
    eps = 1e-6
    J = np.zeros((E, 6, 12))
    for a in range(6):
        d = np.zeros((E, 6))
        d[:, a] = eps
        J[:, :, a] = (residual(mul(Xi, v2t(d)), Xj, Zinv) - residual(mul(Xi, v2t(-d)), Xj, Zinv)) / (2 * eps)
        J[:, :, 6 + a] = (residual(Xi, mul(Xj, v2t(d)), Zinv) - residual(Xi, mul(Xj, v2t(-d)), Zinv)) / (2 * eps)
    H = np.einsum("nra,nrb->nab", J, J)
    bb = np.einsum("nra,nr->na", J, r0)
    idx = np.concatenate([6 * ij[:, :1] + np.arange(6), 6 * ij[:, 1:] + np.arange(6)], axis=1)
    rows, cols = np.repeat(idx, 12, axis=1).ravel(), np.tile(idx, (1, 12)).ravel()
    A = sp.coo_matrix((H.ravel(), (rows, cols)), shape=(6 * V, 6 * V)).tocsr()
    b = np.zeros(6 * V)
    np.add.at(b, idx.ravel(), bb.ravel())
    keep = np.arange(6, 6 * V)  # pose 0 is Fixed (multi_graph_slam_impl.cpp:86)
    A = A[keep][:, keep].tocsc()
    t0 = time.time()
    lu = spl.splu(A, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    dx = np.zeros(6 * V)
    dx[6:] = lu.solve(-b[keep])
    print("factorised and solved in %.0f s, fill %d non-zeros, residual %.1e" %
          (time.time() - t0, lu.L.nnz + lu.U.nnz, np.linalg.norm(A @ dx[6:] + b[keep]) / np.linalg.norm(b[keep])), flush=True)
    after = mul(X, v2t(dx.reshape(V, 6)))
    return chi, dx, after


def main():
    V, E = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (50_000, 200_000)
    chi, dx, after = gauss_newton_step(V, E, 5000)
    name = "posegraph_golden_c5.npz" if (V, E) == (50_000, 200_000) else "posegraph_golden_%d.npz" % V
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), name)
    np.savez_compressed(path, chi0=chi, poses_after_1=after.astype(np.float32), max_abs_dx=np.max(np.abs(dx)))
    print("wrote", path, os.path.getsize(path), "chi0", chi, "max |dx|", np.max(np.abs(dx)))


if __name__ == "__main__":
    main()
