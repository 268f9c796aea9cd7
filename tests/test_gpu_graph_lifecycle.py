"""-m gpu: the incremental pose-graph interface (srrg2_posegraph_add_variable / add_factor / set_factor_enabled /
remove_factor) against the one-shot interface and the CPU oracle, driven through the MultiGraphSLAM_ lifecycle mirror
(S/system/multi_graph_slam_impl.cpp:52-90, :227-297, :300-317)."""
import numpy as np
import pytest

from srrg2_slam_interfaces_amd import _abi as abi
from srrg2_slam_interfaces_amd import graph_slam
from srrg2_slam_interfaces_amd import posegraph as pgm
from srrg2_slam_interfaces_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _rel(Xi, Xj):
    A = np.vstack([Xi, [0, 0, 0, 1]]).astype(np.float64)
    B = np.vstack([Xj, [0, 0, 0, 1]]).astype(np.float64)
    return (np.linalg.inv(A) @ B)[:3].astype(np.float32)


def test_lifecycle_matches_one_shot_graph_and_oracle(oracle, product):
    kind = abi.SE3_QUAT_RIGHT
    g = syn.pose_graph_3d(V=400, E=1200, seed=77)  # ground truth, noisy initial poses, odometry chain + closures
    poses0, ij, Z = g["poses_init"], g["ij"], g["Z"]
    V = poses0.shape[0]
    chain = [e for e in range(len(ij)) if ij[e][1] == ij[e][0] + 1]
    closures = [e for e in range(len(ij)) if e not in set(chain)]
    assert len(chain) == V - 1 and len(closures) > 100
    inc = product.PoseGraph(kind, 0)
    life = graph_slam.GraphSLAMLifecycle(inc)
    by_pair = {tuple(ij[e]): e for e in chain}
    # local maps arrive one by one; every 100 maps a batch of closures is detected, validated, optimised
    params = pgm.PoseGraphParams(5, 100, 1e-8, 0.0)
    rejected, accepted_edges, order = set(), [], []
    pending = sorted(closures, key=lambda e: max(ij[e]))
    for v in range(V):
        Zodo = Z[by_pair[(v - 1, v)]] if v > 0 else syn.identity(3)
        life.make_new_map(poses0[v], Zodo)
        if v > 0:
            order.append(by_pair[(v - 1, v)])
        if v % 100 == 99 or v == V - 1:
            batch = [e for e in pending if max(ij[e]) <= v]
            pending = [e for e in pending if max(ij[e]) > v]
            verdict = lambda ids: [graph_slam.REJECTED if k % 5 == 0 else graph_slam.ACCEPTED for k in range(len(ids))]
            before = inc.size()
            acc = life.loop_validate([(int(ij[e][0]), int(ij[e][1]), Z[e], None) for e in batch], verdict)
            order.extend(batch)
            rejected.update(e for k, e in enumerate(batch) if k % 5 == 0)
            assert len(acc) == len(batch) - len([k for k in range(len(batch)) if k % 5 == 0])
            after = inc.size()
            assert after[1] - before[1] == len(acc) and after[0] == v + 1
            assert life.num_valid_closures == len(acc)
    stats_inc = life.optimize(params)
    assert len(stats_inc) == 5
    # the same graph in one shot (same factor order), rejected closures disabled: product and oracle
    ij1 = np.array([ij[e] for e in order], np.int32)
    Z1 = np.array([Z[e] for e in order], np.float32)
    en1 = np.array([0 if e in rejected else 1 for e in order], np.uint8)
    fixed = np.zeros(V, np.uint8); fixed[0] = 1
    one = product.PoseGraph(kind, 0)
    one.set_graph(poses0, ij1, Z1, fixed_mask=fixed, enabled=en1)
    stats_one = one.solve(params)
    assert [s["chi"] for s in stats_inc] == [s["chi"] for s in stats_one]
    assert inc.poses().tobytes() == one.poses().tobytes()
    ref = oracle.OraclePoseGraph(kind)
    ref.set_graph(poses0, ij1, Z1, fixed_mask=fixed, enabled=en1)
    # (the oracle's block-Jacobi PCG needs far more than 100 iterations for what the product's multigrid-preconditioned
    # PCG does in 100: give it the budget to converge to the same tolerance)
    stats_ref = ref.solve(pgm.PoseGraphParams(5, 5000, 1e-8, 0.0))
    assert np.max(np.abs(ref.poses() - inc.poses())) < 1e-4
    assert abs(stats_ref[-1]["chi"] - stats_inc[-1]["chi"]) <= 1e-3 * max(1.0, stats_ref[-1]["chi"])
    assert stats_inc[-1]["chi"] < 0.05 * stats_inc[0]["chi"]
    # misuse
    with pytest.raises(RuntimeError):
        inc.add_factor(0, V, Z[0])
    removed = order.index(next(iter(rejected)))
    with pytest.raises(RuntimeError):
        inc.set_factor_enabled(removed, True)
    # no valid closure -> optimize() is a no-op (multi_graph_slam_impl.cpp:302-304)
    life.loop_validate([])
    assert life.optimize(params) == []


def test_appended_leaves_are_eliminated_not_rebuilt(oracle, product):
    """makeNewMap appends one variable and one factor per local map (multi_graph_slam_impl.cpp:52-90).  Such variables are leaves:
    the solve eliminates them exactly (Schur complement; the multigrid hierarchy of the graph before them stays valid as it is) and
    back-substitutes -- a chain of three new maps and one more hanging off an old pose, each starting metres away from where its
    factor wants it.  Same poses as the same graph solved in one shot from the same start and as the oracle's; no structure build.
    A closure between two old poses rebuilds, as before."""
    kind = abi.SE3_QUAT_RIGHT
    g = syn.pose_graph_3d(V=3000, E=10000, seed=91)
    poses0, ij, Z = g["poses_init"], g["ij"], g["Z"]
    V, E = poses0.shape[0], ij.shape[0]
    params = pgm.PoseGraphParams(6, 300, 1e-8, 0.0)
    inc = product.PoseGraph(kind, 0)
    inc.set_graph(poses0, ij, Z)
    inc.solve(params)
    assert inc.structure_info() == (1, 0)
    start = inc.poses().copy()
    rng = np.random.RandomState(5)
    new_pose, new_ij, new_Z = [], [], []
    for parent in (V - 1, V, V + 1, 1234):  # a chain V-1 -> V -> V+1 -> V+2, and a leaf on pose 1234
        t = rng.uniform(-0.5, 0.5, 3)
        Zk = syn.se3(t, rng.uniform(-0.2, 0.2, 3)).astype(np.float32)
        # (metres and a third of a radian away from parent * Z, where the factor wants it)
        where = list(start) + new_pose
        far = syn.se3_mul(syn.se3_mul(where[parent], Zk), syn.se3(rng.uniform(-3, 3, 3), rng.uniform(-0.3, 0.3, 3))).astype(np.float32)
        vid = inc.add_variable(far)
        # (both orientations of the new factor: the new pose as its second and as its first endpoint)
        if len(new_pose) % 2 == 0:
            inc.add_factor(parent, vid, Zk)
            new_ij.append((parent, vid)); new_Z.append(Zk)
        else:
            Zinv = np.linalg.inv(np.vstack([Zk, [0, 0, 0, 1]]).astype(np.float64))[:3].astype(np.float32)
            inc.add_factor(vid, parent, Zinv)
            new_ij.append((vid, parent)); new_Z.append(Zinv)
        new_pose.append(far)
    st = inc.solve(params)
    assert inc.structure_info() == (1, 4), inc.structure_info()
    assert all(s_["solver_status"] == 0 and s_["pcg_iterations"] < 300 for s_ in st)
    # the same graph in one shot from the same start
    P1 = np.concatenate([start, np.array(new_pose, np.float32)], axis=0)
    ij1 = np.concatenate([ij, np.array(new_ij, np.int32)], axis=0)
    Z1 = np.concatenate([Z, np.array(new_Z, np.float32)], axis=0)
    one = product.PoseGraph(kind, 0)
    one.set_graph(P1, ij1, Z1)
    st1 = one.solve(params)
    assert one.structure_info() == (1, 0)
    assert np.max(np.abs(one.poses() - inc.poses())) < 2e-4
    assert abs(st1[-1]["chi"] - st[-1]["chi"]) <= 1e-3 * max(1.0, st1[-1]["chi"])
    # every new pose sits where its factor puts it (a leaf's factor ends with zero residual)
    Xf = inc.poses()
    for (i, j), Zk in zip(new_ij, new_Z):
        assert np.max(np.abs(_rel(Xf[i], Xf[j]) - Zk)) < 1e-4
    ref = oracle.OraclePoseGraph(kind)
    ref.set_graph(P1, ij1, Z1)
    ref.solve(pgm.PoseGraphParams(6, 8000, 1e-8, 0.0))
    assert np.max(np.abs(ref.poses() - inc.poses())) < 5e-4
    # a closure between two old poses: the structure is rebuilt, nothing is eliminated any more
    inc.add_factor(10, 2000, _rel(Xf[10], Xf[2000]))
    inc.solve(params)
    assert inc.structure_info() == (2, 0)


def test_appended_leaves_se2_and_a_leaf_on_the_fixed_pose(oracle, product):
    """the same through SE(2), with one leaf hanging off the FIXED pose 0 (an identity row: nothing is folded into it, the leaf's
    step is its own) and one off a free pose; the spanning-tree geometry of the matching starts at that fixed pose"""
    kind = abi.SE2_RIGHT
    g = syn.pose_graph_2d(V=1500, E=4000, seed=5301)
    poses0, ij, Z = g["poses_init"], g["ij"], g["Z"]
    V = poses0.shape[0]
    params = pgm.PoseGraphParams(6, 300, 1e-8, 0.0)
    inc = product.PoseGraph(kind, 0)
    inc.set_graph(poses0, ij, Z)
    inc.solve(params)
    assert inc.structure_info() == (1, 0)
    start = inc.poses().copy()
    new_pose, new_ij, new_Z = [], [], []
    for n, parent in enumerate((0, 700)):
        Zk = syn.se2(0.4 - 0.1 * n, -0.2, 0.3 + 0.2 * n).astype(np.float32)
        far = (start[parent].astype(np.float64) @ Zk @ syn.se2(1.5, -2.0, 0.25)).astype(np.float32)
        vid = inc.add_variable(far)
        inc.add_factor(parent, vid, Zk)
        new_pose.append(far); new_ij.append((parent, vid)); new_Z.append(Zk)
    st = inc.solve(params)
    assert inc.structure_info() == (1, 2)
    assert all(s_["solver_status"] == 0 for s_ in st)
    P1 = np.concatenate([start, np.array(new_pose, np.float32)], axis=0)
    ij1 = np.concatenate([ij, np.array(new_ij, np.int32)], axis=0)
    Z1 = np.concatenate([Z, np.array(new_Z, np.float32)], axis=0)
    one = product.PoseGraph(kind, 0)
    one.set_graph(P1, ij1, Z1)
    one.solve(params)
    assert np.max(np.abs(one.poses() - inc.poses())) < 2e-4
    Xf = inc.poses().astype(np.float64)
    for (i, j), Zk in zip(new_ij, new_Z):
        assert np.max(np.abs(np.linalg.inv(Xf[i]) @ Xf[j] - Zk)) < 1e-4
    ref = oracle.OraclePoseGraph(kind)
    ref.set_graph(P1, ij1, Z1)
    ref.solve(pgm.PoseGraphParams(6, 8000, 1e-8, 0.0))
    assert np.max(np.abs(ref.poses() - inc.poses())) < 5e-4


def test_structure_geometry_without_a_fixed_variable_and_with_two_components(oracle, product):
    """the matching's spanning tree (build_hierarchy): a graph of two components, only one of which holds the Fixed pose -- the other
    one's tree starts at its first pose with the pose it has; the free component is a null space of H that the coarsest solve leaves
    alone (its vanished pivots), the anchored component converges like the graph on its own"""
    kind = abi.SE3_QUAT_RIGHT
    a = syn.pose_graph_3d(V=1200, E=4000, seed=611)
    b = syn.pose_graph_3d(V=800, E=2600, seed=612)
    Va = a["poses_init"].shape[0]
    poses = np.concatenate([a["poses_init"], b["poses_init"]], axis=0)
    ij = np.concatenate([a["ij"], b["ij"] + Va], axis=0).astype(np.int32)
    Z = np.concatenate([a["Z"], b["Z"]], axis=0)
    params = pgm.PoseGraphParams(6, 400, 1e-7, 1e-6)  # (a little damping: the unanchored component is singular without it)
    both = product.PoseGraph(kind, 0)
    both.set_graph(poses, ij, Z)
    st = both.solve(params)
    assert all(s_["solver_status"] == 0 for s_ in st)
    alone = product.PoseGraph(kind, 0)
    alone.set_graph(a["poses_init"], a["ij"], a["Z"])
    alone.solve(params)
    assert np.max(np.abs(both.poses()[:Va] - alone.poses())) < 5e-4
    assert st[-1]["chi"] < 0.05 * st[0]["chi"]
