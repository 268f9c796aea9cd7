#!/usr/bin/env python
"""A laser tracker's aligner: one nearest-neighbour cue slice (2-D scan, SE(2) point-to-point) NEXT TO an odometry prior slice --
the control steps of such an aligner are launches (no fused control steps).  ms per compute(), steady state and on a new fixed
scan.   usage: python tools/bench_prior_cue.py [beams ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import srrg2_slam_interfaces_amd as pkg
from srrg2_slam_interfaces_amd import _abi as abi, synthetic as syn

with_prior = "--no-prior" not in sys.argv
if "--3d" in sys.argv:  # a 3-D tracker: 100 000-point SE(3) point-to-plane cue slice (C2) next to a motion-model prior
    n = [int(x) for x in sys.argv[1:] if not x.startswith("--")] or [100000]
    for npts in n:
        d = syn.cloud_pair_3d(n=npts, seed=77)
        al = pkg.MultiAligner(abi.SE3_QUAT_RIGHT)
        al.set_params(max_iterations=10, min_num_inliers=10)
        c = abi.default_slice_config(abi.SE3_QUAT_RIGHT)
        c.kind, c.finder, c.finder_max_distance = abi.SLICE_P2PLANE, abi.FINDER_NN_GATED, 0.25
        c.robustifier, c.robustifier_chi_threshold = abi.ROBUST_CAUCHY, 0.05
        si = al.add_slice(c)
        if with_prior:
            p = abi.default_slice_config(abi.SE3_QUAT_RIGHT)
            p.kind, p.finder, p.prior_sets_initial_guess = abi.SLICE_PRIOR, abi.FINDER_NONE, 0
            for i, v in enumerate([10.0, 10.0, 10.0, 100.0, 100.0, 100.0]):
                p.prior_information_diag[i] = v
            pi = al.add_slice(p)
        al.set_fixed(si, d["fixed"], d["fixed_normals"])
        al.set_moving(si, d["moving"], d["moving_normals"])
        if with_prior:
            al.set_prior_measurement(pi, syn.identity(3).astype(np.float32))
        steady, fresh = [], []
        for rep in range(60):
            al.set_moving_in_fixed(syn.identity(3))
            t0 = time.perf_counter()
            st = al.compute()
            steady.append(time.perf_counter() - t0)
        for rep in range(40):
            al.set_fixed(si, d["fixed"], d["fixed_normals"])
            al.set_moving_in_fixed(syn.identity(3))
            t0 = time.perf_counter()
            al.compute()
            fresh.append(time.perf_counter() - t0)
        print("%7d 3-D points" % npts + (" + prior" if with_prior else " (cue slice only)") + ": steady %.4f ms per compute(), on a new fixed cloud %.4f ms, status %d"
              % (1e3 * float(np.median(steady[5:])), 1e3 * float(np.median(fresh[5:])), st), flush=True)
    sys.exit(0)
for beams in [int(x) for x in sys.argv[1:] if not x.startswith("--")] or [360, 1000, 3000]:
    d = syn.scan_pair_2d(beams=beams, sigma=0.01, seed=1234)
    al = pkg.MultiAligner(abi.SE2_RIGHT)
    al.set_params(max_iterations=10, min_num_inliers=10)
    c = abi.default_slice_config(abi.SE2_RIGHT)
    c.kind, c.finder, c.finder_max_distance = abi.SLICE_P2P, abi.FINDER_NN_GATED, 0.5
    c.robustifier, c.robustifier_chi_threshold = abi.ROBUST_CAUCHY, 0.05
    si = al.add_slice(c)
    if with_prior:
        p = abi.default_slice_config(abi.SE2_RIGHT)
        p.kind, p.finder, p.prior_sets_initial_guess = abi.SLICE_PRIOR, abi.FINDER_NONE, 0
        for i, v in enumerate([10.0, 10.0, 100.0]):
            p.prior_information_diag[i] = v
        pi = al.add_slice(p)
    al.set_fixed(si, d["fixed"])
    al.set_moving(si, d["moving"])
    if with_prior:
        al.set_prior_measurement(pi, syn.identity(2).astype(np.float32))
    steady, fresh = [], []
    for rep in range(40):
        al.set_moving_in_fixed(syn.identity(2))
        t0 = time.perf_counter()
        st = al.compute()
        steady.append(time.perf_counter() - t0)
    for rep in range(30):
        al.set_fixed(si, d["fixed"])
        al.set_moving_in_fixed(syn.identity(2))
        t0 = time.perf_counter()
        al.compute()
        fresh.append(time.perf_counter() - t0)
    print("%5d beams" % beams + (" + prior" if with_prior else " (cue slice only)") + ": steady %.4f ms per compute(), on a new fixed scan %.4f ms, status %d"
          % (1e3 * float(np.median(steady[5:])), 1e3 * float(np.median(fresh[5:])), st), flush=True)
