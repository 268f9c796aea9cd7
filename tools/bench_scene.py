#!/usr/bin/env python
"""Timing of the scene steps with the clouds resident in HBM:
ball clip of an N-point scene, correspondence-based merge from an aligner run.  Prints one JSON line.
Algorithmic bytes: clip = 16 N (flag pass) + 16 N (+16 N normals of the kept) + 36 kept written;
merge = 9 Nm (aligner per-point outputs) + 16 Nm (sorted moving, .w) + per correspondence 4 (gidx) + 32 read + 32 written."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

import srrg2_slam_interfaces_amd as pkg
from srrg2_slam_interfaces_amd import _abi as abi
from srrg2_slam_interfaces_amd import mapping
from srrg2_slam_interfaces_amd import synthetic as syn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=2_000_000)
    ap.add_argument("--frame-points", type=int, default=100_000)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    b = pkg.scene_binding(0)
    rng = np.random.default_rng(1)
    P, N = syn.scene_3d(args.points, 500)
    full, clipped, meas = mapping.Scene(b, 3), mapping.Scene(b, 3), mapping.Scene(b, 3)
    full.set(P.astype(np.float32), N.astype(np.float32))
    cl = mapping.SceneClipperBall(b, range_max=8.0)
    cl.set_full_scene(full); cl.set_clipped_scene_in_robot(clipped); cl.set_robot_in_local_map(syn.identity(3))
    cl.compute()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        cl.compute()
    t_clip = (time.perf_counter() - t0) / args.reps
    kept = clipped.size()
    clip_bytes = 32 * args.points + 16 * kept + 36 * kept
    # merge: align a fresh 100k frame against the first 100k scene points, merge it in
    d = syn.cloud_pair_3d(n=args.frame_points, seed=2000)
    scene = mapping.Scene(b, 3)
    scene.set(d["moving"], d["moving_normals"])
    meas.set(d["fixed"], d["fixed_normals"])
    cl2 = mapping.SceneClipperBall(b, range_max=1e3)
    clipped2 = mapping.Scene(b, 3)
    cl2.set_full_scene(scene); cl2.set_clipped_scene_in_robot(clipped2); cl2.set_robot_in_local_map(syn.identity(3))
    cl2.compute()
    al = pkg.MultiAligner(abi.SE3_QUAT_RIGHT, 0)
    c = abi.default_slice_config(abi.SE3_QUAT_RIGHT)
    c.kind = abi.SLICE_P2PLANE; c.finder_max_distance = 0.25; c.robustifier = abi.ROBUST_CAUCHY
    c.robustifier_chi_threshold = 0.05
    si = al.add_slice(c)
    cp, cn, n = clipped2.device_arrays()
    al.set_cloud_device("set_moving", si, cp, 16, cn, 16, n)
    mp, mn, m = meas.device_arrays()
    al.set_cloud_device("set_fixed", si, mp, 16, mn, 16, m)
    al.set_moving_in_fixed(syn.identity(3))
    al.compute()
    X = al.moving_in_fixed()
    Xi = np.linalg.inv(np.vstack([X, [0, 0, 0, 1]]).astype(np.float64))[:3].astype(np.float32)
    mg = mapping.MergerCorrespondenceHomo(b, mapping.MergerParams(50.0, 0.01, 10 ** 9))
    mg.set_scene(scene); mg.set_measurement(meas); mg.set_measurement_in_scene(Xi)
    times = []
    res = None
    for _ in range(args.reps):
        scene.set(d["moving"], d["moving_normals"])  # (untimed) restore the scene
        t0 = time.perf_counter()
        res = mg.compute_from_aligner(al, si, clipped2)
        times.append(time.perf_counter() - t0)
    t_merge = float(np.median(times))
    nm = args.frame_points
    merge_bytes = 25 * nm + res["num_correspondences"] * 68 + 17 * nm + res["num_added"] * 64
    print(json.dumps({
        "clip": {"points": args.points, "kept": kept, "ms": 1e3 * t_clip, "algorithmic_GBps": clip_bytes / t_clip / 1e9,
                 "frac_of_8TBps": clip_bytes / t_clip / 8e12},
        "merge_from_aligner": {"scene": nm, "measurement": nm, **res, "ms": 1e3 * t_merge,
                               "algorithmic_GBps": merge_bytes / t_merge / 1e9, "frac_of_8TBps": merge_bytes / t_merge / 8e12},
    }))


if __name__ == "__main__":
    main()
