"""CPU: the oracle's pose-graph solve against finite differences, against its own dense
Cholesky solve, and against SciPy's sparse direct solver (golden fixture made in this container)."""
import os

import numpy as np
import pytest

from srrg2_slam_interfaces_amd import _abi as abi
from srrg2_slam_interfaces_amd import posegraph as pgm
from srrg2_slam_interfaces_amd import synthetic as syn


def _graph(oracle, kind, g, **kw):
    pg = oracle.OraclePoseGraph(kind)
    pg.set_graph(g["poses_init"], g["ij"], g["Z"], **kw)
    return pg


@pytest.mark.parametrize("kind", [abi.SE3_QUAT_RIGHT, abi.SE2_RIGHT])
def test_edge_jacobians_match_finite_differences(oracle, kind):
    g = syn.pose_graph_3d(V=30, E=60, seed=7) if kind == abi.SE3_QUAT_RIGHT else syn.pose_graph_2d(V=30, E=60, seed=7)
    pg = _graph(oracle, kind, g)
    D = pg.D
    poses0 = g["poses_init"].copy()
    eps = 1e-3
    for e in (0, 5, 40, pg.E - 1):
        err, Ji, Jj = pg.edge(e)
        i, j = g["ij"][e]
        for which, J in ((i, Ji), (j, Jj)):
            Jn = np.zeros((D, D))
            for a in range(D):
                dx = np.zeros(D)
                dx[a] = eps
                outs = []
                for sgn in (1, -1):
                    P = poses0.copy()
                    P[which] = oracle.box_plus(kind, poses0[which], sgn * dx)
                    pg.set_graph(P, g["ij"], g["Z"])
                    outs.append(pg.edge(e)[0])
                Jn[:, a] = (outs[0] - outs[1]) / (2 * eps)
            pg.set_graph(poses0, g["ij"], g["Z"])
            assert np.max(np.abs(J - Jn)) < 5e-4, (e, which, J, Jn)


@pytest.mark.parametrize("kind", [abi.SE3_QUAT_RIGHT, abi.SE2_RIGHT])
def test_pcg_equals_dense_cholesky_and_converges(oracle, kind):
    g = syn.pose_graph_3d(V=120, E=400, seed=8) if kind == abi.SE3_QUAT_RIGHT else syn.pose_graph_2d(V=150, E=400)
    p = pgm.default_params()
    p.pcg_tolerance = 1e-10
    p.pcg_max_iterations = 2000
    a = _graph(oracle, kind, g)
    b = _graph(oracle, kind, g)
    b.set_direct(True)
    chi0 = a.chi()
    sa, sb = a.solve(p), b.solve(p)
    assert len(sa) == len(sb) == 10 and all(s["solver_status"] == 0 for s in sa + sb)
    assert np.max(np.abs(a.poses() - b.poses())) < 2e-6          # float32 poses, same fixed point
    assert abs(sa[0]["chi"] - chi0) / chi0 < 1e-5 and abs(sa[0]["chi"] - sb[0]["chi"]) < 1e-3 * chi0
    assert sa[-1]["chi"] < 0.2 * sa[0]["chi"]                    # odometry drift is corrected
    assert a.chi() <= sa[-1]["chi"] * 1.001
    # the optimum is close to the ground truth (noise level), the initial guess is not
    err_init = np.max(np.abs(g["poses_init"] - g["poses_gt"]))
    err_opt = np.max(np.abs(a.poses() - g["poses_gt"]))
    assert err_opt < 0.3 * err_init
    assert np.array_equal(a.poses()[0], g["poses_init"][0])      # pose 0 is Fixed (multi_graph_slam_impl.cpp:86)


def test_disabled_factors_are_skipped_and_fixed_mask(oracle):
    kind = abi.SE3_QUAT_RIGHT
    g = syn.pose_graph_3d(V=60, E=200, seed=9)
    en = np.ones(g["ij"].shape[0], np.uint8)
    en[59:] = 0  # closures created disabled (loop_closure.h:71): only odometry left
    pg = _graph(oracle, kind, g, enabled=en)
    st = pg.solve()
    assert st[0]["num_factors"] == 59
    # odometry-only chain with the integrated guess is already optimal: chi == 0, poses unchanged
    assert st[0]["chi"] < 1e-8 and np.max(np.abs(pg.poses() - g["poses_init"])) < 1e-6
    pg.set_enabled(np.ones_like(en))
    st = pg.solve()
    assert st[0]["num_factors"] == g["ij"].shape[0] and st[0]["chi"] > 1e-4
    fm = np.zeros(60, np.uint8)
    fm[[0, 17]] = 1
    pg2 = _graph(oracle, kind, g, fixed_mask=fm)
    pg2.solve()
    assert np.array_equal(pg2.poses()[17], g["poses_init"][17])


def test_matches_scipy_sparse_golden(oracle):
    """tests/golden/posegraph_golden.npz: first Gauss-Newton increment of a 100-pose graph solved with
    scipy.sparse.linalg.spsolve on a system assembled by an independent numpy implementation."""
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "posegraph_golden.npz"))
    g = syn.pose_graph_3d(V=100, E=300, seed=11)
    pg = _graph(oracle, abi.SE3_QUAT_RIGHT, g)
    pg.set_direct(True)
    p = pgm.default_params()
    p.max_iterations = 1
    st = pg.solve(p)
    assert abs(st[0]["chi"] - float(G["chi0"])) / float(G["chi0"]) < 1e-4
    assert np.max(np.abs(pg.poses() - G["poses_after_1"])) < 1e-5


def test_oracle_matches_scipy_direct_solve_2k(oracle):
    """the 2 048-pose fixture of SciPy's sparse direct solver (tests/golden/make_posegraph_golden.py: independent
    assembly, numerical Jacobians): the oracle's linearisation and PCG reproduce its Gauss-Newton step"""
    import os

    from srrg2_slam_interfaces_amd import posegraph as pgm

    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "posegraph_golden_2k.npz"))
    g = syn.pose_graph_3d(V=2048, E=8192, seed=12)
    p = pgm.default_params()
    p.pcg_tolerance, p.pcg_max_iterations, p.max_iterations = 1e-10, 20000, 1
    pg = oracle.OraclePoseGraph(abi.SE3_QUAT_RIGHT)
    pg.set_graph(g["poses_init"], g["ij"], g["Z"])
    st = pg.solve(p)
    assert st[0]["solver_status"] == 0 and st[0]["pcg_iterations"] < 20000
    assert abs(st[0]["chi"] - float(G["chi0"])) / float(G["chi0"]) < 1e-6
    assert np.max(np.abs(pg.poses() - G["poses_after_1"])) < 2e-5


def test_se2_oracle_matches_scipy_direct_solve(oracle):
    """SE(2), what srrg2_laser_slam_2d optimises (S/mapping/local_map.h:64): 1 500 poses / ~4 500 factors, two Gauss-Newton
    steps against the committed result of SciPy's sparse DIRECT solver on an independently assembled system
    (tests/golden/make_posegraph_golden_se2.py: numpy residual, central-difference Jacobians) -- VERDICT r3 #6"""
    import os

    from srrg2_slam_interfaces_amd import posegraph as pgm

    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "posegraph_golden_se2.npz"))
    g = syn.pose_graph_2d(V=1500, E=4500, seed=5300)
    p = pgm.default_params()
    p.pcg_tolerance, p.pcg_max_iterations = 1e-10, 40000
    for its, key, chi_key in ((1, "poses_after_1", "chi0"), (2, "poses_after_2", "chi1")):
        pg = oracle.OraclePoseGraph(abi.SE2_RIGHT)
        pg.set_graph(g["poses_init"], g["ij"], g["Z"])
        p.max_iterations = its
        st = pg.solve(p)
        assert all(s["solver_status"] == 0 and s["pcg_iterations"] < 40000 for s in st)
        assert abs(st[-1]["chi"] - float(G[chi_key])) / float(G[chi_key]) < 1e-4
        assert np.max(np.abs(pg.poses() - G[key])) < 2e-5 * max(1.0, float(G["max_abs_dx"][0]))
