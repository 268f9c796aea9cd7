"""Driven by tests/test_oracle_sanitizers.py inside a process that preloads libasan: the CPU oracle (a build of oracle/*.c
with -fsanitize=address,undefined, SRRG2_ORACLE_LIB) through its main entry points -- NN aligner with the inlier-only run
and pruning, a batch, the projective two-slice aligner, the SE(2) aligner with a prior slice, the pose graph."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from oracle import pyoracle  # noqa: E402
from srrg2_slam_interfaces_amd import _abi as abi  # noqa: E402
from srrg2_slam_interfaces_amd import posegraph as pgm  # noqa: E402
from srrg2_slam_interfaces_amd import synthetic as syn  # noqa: E402


def cue(kind, slice_kind, gate):
    c = abi.default_slice_config(kind)
    c.kind, c.finder, c.finder_max_distance = slice_kind, abi.FINDER_NN_GATED, gate
    c.robustifier, c.robustifier_chi_threshold, c.finder_normal_cos = abi.ROBUST_CAUCHY, 0.05, 0.8
    return c


# SE(3) point-to-plane, inlier-only run + pruning, termination criterion
d = syn.cloud_pair_3d(n=4000, seed=5)
al = pyoracle.OracleAligner(abi.SE3_QUAT_RIGHT)
al.set_params(max_iterations=8, min_num_inliers=10, enable_inlier_only_runs=1, keep_only_inlier_correspondences=1)
tp = abi.TerminationParams() if hasattr(abi, "TerminationParams") else None
if tp is not None:
    tp.window_size, tp.num_correspondences_range, tp.num_inliers_range, tp.num_outliers_range, tp.chi_epsilon = 3, 50, 50, 50, 0.5
    al.set_termination_criteria(tp)
si = al.add_slice(cue(abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE, 0.25))
al.set_fixed(si, d["fixed"], d["fixed_normals"])
al.set_moving(si, d["moving"], d["moving_normals"])
al.set_moving_in_fixed(syn.identity(3))
assert al.compute() == abi.SUCCESS
assert len(al.correspondences(0)) > 1000 and len(al.iteration_stats()) >= 2
# a ragged batch (one empty cloud) against the same fixed cloud
probs = syn.batch_3d(K=3, n=1500, seed=77, shared_fixed_group=64)
al2 = pyoracle.OracleAligner(abi.SE3_QUAT_RIGHT)
al2.set_params(max_iterations=4, min_num_inliers=10)
al2.add_slice(cue(abi.SE3_QUAT_RIGHT, abi.SLICE_P2P, 0.3))
al2.set_fixed(0, probs[0]["fixed"], probs[0]["fixed_normals"])
movs = [probs[0]["moving"], probs[1]["moving"][:0], probs[2]["moving"][:700]]
nrms = [probs[0]["moving_normals"], probs[1]["moving_normals"][:0], probs[2]["moving_normals"][:700]]
res = al2.compute_batch(movs, [syn.identity(3)] * 3, nrms)
assert res[0]["status"] == abi.SUCCESS and res[1]["status"] != abi.SUCCESS
# projective finder, two slices
r = syn.rgbd_pair(rows=60, cols=80, seed=3)
al3 = pyoracle.OracleAligner(abi.SE3_QUAT_RIGHT)
al3.set_params(max_iterations=3, min_num_inliers=10)
for sk in (abi.SLICE_P2PLANE, abi.SLICE_REPROJECTION):
    c = abi.default_slice_config(abi.SE3_QUAT_RIGHT)
    c.kind, c.finder, c.finder_max_distance = sk, abi.FINDER_PROJECTIVE, 0.05
    for i, v in enumerate(r["K"].reshape(-1)):
        c.camera_matrix[i] = v
    c.image_rows, c.image_cols, c.depth_min, c.depth_max = r["rows"], r["cols"], r["depth_min"], r["depth_max"]
    s3 = al3.add_slice(c)
    al3.set_fixed(s3, r["fixed"], r["fixed_normals"])
    al3.set_moving(s3, r["moving"], r["moving_normals"])
al3.set_moving_in_fixed(syn.identity(3))
al3.compute()
# SE(2) scan pair with an odometry prior slice
s2 = syn.scan_pair_2d(seed=11)
al4 = pyoracle.OracleAligner(abi.SE2_RIGHT)
al4.set_params(max_iterations=5, min_num_inliers=5)
c = cue(abi.SE2_RIGHT, abi.SLICE_P2P, 0.5)
c.finder_normal_cos = -2.0
al4.add_slice(c)
p = abi.default_slice_config(abi.SE2_RIGHT)
p.kind, p.finder = abi.SLICE_PRIOR, abi.FINDER_NONE
sp = al4.add_slice(p)
al4.set_fixed(0, s2["fixed"], None)
al4.set_moving(0, s2["moving"], None)
al4.set_prior_measurement(sp, syn.identity(2))
al4.set_moving_in_fixed(syn.identity(2))
al4.compute()
# pose graph: SE(3), two Gauss-Newton iterations
g = syn.pose_graph_3d(V=300, E=900, seed=9)
pg = pyoracle.OraclePoseGraph(abi.SE3_QUAT_RIGHT)
pg.set_graph(g["poses_init"], g["ij"], g["Z"])
prm = pgm.default_params()
prm.max_iterations = 2
st = pg.solve(prm)
assert len(st) == 2 and np.isfinite(st[-1]["chi"])
print("oracle sanitizer drive ok")
