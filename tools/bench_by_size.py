#!/usr/bin/env python
"""One SE(3) point-to-plane alignment by cloud size: steady state (clouds bound once, lists built) and a first compute() on a
new fixed cloud.   usage: python tools/bench_by_size.py [n ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import srrg2_slam_interfaces_amd as pkg
from srrg2_slam_interfaces_amd import _abi as abi, synthetic as syn

sizes = [int(x) for x in sys.argv[1:]] or [1000, 3000, 10000, 30000, 60000, 100000, 200000, 400000, 1000000]
for n in sizes:
    d = syn.cloud_pair_3d(n=n, seed=2000, noise_sigma=0.002)
    al = pkg.MultiAligner(abi.SE3_QUAT_RIGHT)
    al.set_params(max_iterations=10, min_num_inliers=10)
    c = abi.default_slice_config(abi.SE3_QUAT_RIGHT)
    c.kind, c.finder, c.finder_max_distance, c.finder_normal_cos = abi.SLICE_P2PLANE, abi.FINDER_NN_GATED, 0.25, 0.8
    c.robustifier, c.robustifier_chi_threshold = abi.ROBUST_CAUCHY, 0.05
    al.add_slice(c)
    al.set_fixed(0, d["fixed"], d["fixed_normals"])
    al.set_moving(0, d["moving"], d["moving_normals"])
    steady, fresh = [], []
    for rep in range(30):
        al.set_moving_in_fixed(syn.identity(3))
        t0 = time.perf_counter()
        st = al.compute()
        steady.append(time.perf_counter() - t0)
    for rep in range(20):
        al.set_fixed(0, d["fixed"], d["fixed_normals"])
        al.set_moving_in_fixed(syn.identity(3))
        t0 = time.perf_counter()
        al.compute()
        fresh.append(time.perf_counter() - t0)
    ms, mf = 1e3 * float(np.median(steady[5:])), 1e3 * float(np.median(fresh[3:]))
    print("%8d points: steady %.4f ms per compute() (%.1f M point-iterations/s), first compute() on a new fixed cloud %.4f ms, status %d"
          % (n, ms, 10 * n / ms / 1e3, mf, st), flush=True)
