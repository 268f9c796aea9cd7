#!/usr/bin/env python
"""A tracker's aligner: one nearest-neighbour cue slice NEXT TO a prior slice (an odometry prior / a motion model: S/instances.cpp:35-38).
2-D: a laser scan, SE(2) point-to-point; --3d: an SE(3) point-to-plane cloud pair (C2's slice).  ms per compute(), steady state (the
fixed cloud kept: lists) and on a new fixed cloud (a tracker's frame).  Since round 6 the control steps of such an aligner are fused
into the pass kernels too (the control wave linearises the prior factor: wave_prior); SRRG2_AMD_FUSED_CONTROL=2 keeps the launches.
usage: python tools/bench_prior_cue.py [--3d] [--no-prior] [points ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import srrg2_slam_interfaces_amd as pkg
from srrg2_slam_interfaces_amd import _abi as abi, synthetic as syn


def run(points, three_d=False, with_prior=True, reps=60, fused_control=None):
    if three_d:
        kind, dim = abi.SE3_QUAT_RIGHT, 3
        d = syn.cloud_pair_3d(n=points, seed=77)
        cue = (abi.SLICE_P2PLANE, 0.25)
        info = [10.0, 10.0, 10.0, 100.0, 100.0, 100.0]
    else:
        kind, dim = abi.SE2_RIGHT, 2
        d = syn.scan_pair_2d(beams=points, sigma=0.01, seed=1234)
        cue = (abi.SLICE_P2P, 0.5)
        info = [10.0, 10.0, 100.0]
    al = pkg.MultiAligner(kind)
    if fused_control is not None:
        al.set_tuning(fused_control=fused_control)
    al.set_params(max_iterations=10, min_num_inliers=10)
    c = abi.default_slice_config(kind)
    c.kind, c.finder, c.finder_max_distance = cue[0], abi.FINDER_NN_GATED, cue[1]
    c.robustifier, c.robustifier_chi_threshold = abi.ROBUST_CAUCHY, 0.05
    si = al.add_slice(c)
    if with_prior:
        p = abi.default_slice_config(kind)
        p.kind, p.finder, p.prior_sets_initial_guess = abi.SLICE_PRIOR, abi.FINDER_NONE, 0
        for i, v in enumerate(info):
            p.prior_information_diag[i] = v
        pi = al.add_slice(p)
    al.set_fixed(si, d["fixed"], d.get("fixed_normals"))
    al.set_moving(si, d["moving"], d.get("moving_normals"))
    if with_prior:
        al.set_prior_measurement(pi, syn.identity(dim).astype(np.float32))
    steady, fresh = [], []
    st = -1
    for rep in range(reps):
        al.set_moving_in_fixed(syn.identity(dim))
        t0 = time.perf_counter()
        st = al.compute()
        steady.append(time.perf_counter() - t0)
    for rep in range(max(reps * 2 // 3, 8)):
        al.set_fixed(si, d["fixed"], d.get("fixed_normals"))
        al.set_moving_in_fixed(syn.identity(dim))
        t0 = time.perf_counter()
        al.compute()
        fresh.append(time.perf_counter() - t0)
    return {"points": points, "dim": dim, "prior": bool(with_prior), "steady_ms": 1e3 * float(np.median(steady[5:])),
            "new_fixed_cloud_ms": 1e3 * float(np.median(fresh[5:])), "status": int(st)}


if __name__ == "__main__":
    three_d = "--3d" in sys.argv
    with_prior = "--no-prior" not in sys.argv
    sizes = [int(x) for x in sys.argv[1:] if not x.startswith("--")] or ([100000] if three_d else [360, 1000, 3000])
    for n in sizes:
        r = run(n, three_d, with_prior)
        print("%7d %s" % (n, "3-D points" if three_d else "beams") + (" + prior" if with_prior else " (cue slice only)") +
              ": steady %.4f ms per compute(), on a new fixed cloud %.4f ms, status %d" % (r["steady_ms"], r["new_fixed_cloud_ms"], r["status"]),
              flush=True)
