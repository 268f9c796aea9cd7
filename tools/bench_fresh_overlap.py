#!/usr/bin/env python
"""A tracker-like first compute() on a NEW fixed cloud (no lists) at full and at 60 % overlap: ms per compute().
  usage: python tools/bench_fresh_overlap.py [n]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import srrg2_slam_interfaces_amd as pkg
from srrg2_slam_interfaces_amd import _abi as abi, synthetic as syn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
for frac in (1.0, 0.6):
    d = syn.cloud_pair_3d(n=n, seed=2600, t=(0.04, -0.03, 0.02), rpy_deg=(0.8, -1.0, 1.2))
    d = {k: v.copy() for k, v in d.items()}
    if frac < 1.0:
        keep = d["fixed"][:, 0] <= np.quantile(d["fixed"][:, 0], frac)
        d["fixed"], d["fixed_normals"] = d["fixed"][keep], d["fixed_normals"][keep]
    al = pkg.MultiAligner(abi.SE3_QUAT_RIGHT)
    al.set_params(max_iterations=10, min_num_inliers=10)
    c = abi.default_slice_config(abi.SE3_QUAT_RIGHT)
    c.kind, c.finder, c.finder_max_distance, c.finder_normal_cos = abi.SLICE_P2PLANE, abi.FINDER_NN_GATED, 0.25, 0.8
    c.robustifier, c.robustifier_chi_threshold = abi.ROBUST_CAUCHY, 0.05
    al.add_slice(c)
    al.set_moving(0, d["moving"], d["moving_normals"])
    ts = []
    for rep in range(40):
        al.set_fixed(0, d["fixed"], d["fixed_normals"])
        al.set_moving_in_fixed(syn.identity(3))
        t0 = time.perf_counter()
        st = al.compute()
        ts.append(time.perf_counter() - t0)
    print("overlap %.0f %%: %.4f ms per first compute() (status %d, %d correspondences)" % (
        100 * frac, 1e3 * float(np.median(ts[5:])), st, al.iteration_stats()[-1]["num_correspondences"]))
