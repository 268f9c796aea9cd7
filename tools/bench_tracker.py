#!/usr/bin/env python
"""Frame rate of the tracker cycle with everything resident in HBM: clip the local map around the last estimate, align the
new measurement against it (set_fixed builds the grid, set_moving sorts the clipped scene), merge.  One JSON line with the
per-stage milliseconds (host wall clock around blocking C-ABI calls)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import srrg2_slam_interfaces_amd as pkg  # noqa: E402
from srrg2_slam_interfaces_amd import _abi as abi  # noqa: E402
from srrg2_slam_interfaces_amd import mapping  # noqa: E402
from srrg2_slam_interfaces_amd import synthetic as syn  # noqa: E402


# the scene arrays stay as they are until compute() has returned (the clip of the next frame and the next measurement come after
# it): SRRG2_MEM_DEVICE_KEPT -- set_moving / set_fixed do not wait for their ingest (SRRG2_TRACKER_KEPT=0: they do)
KEPT = os.environ.get("SRRG2_TRACKER_KEPT", "1") != "0"
FIXED_FIRST = os.environ.get("SRRG2_TRACKER_FIXED_FIRST", "0") != "0"  # 1: set_fixed before set_moving, as MultiTrackerBase_::align does (the same frame time: set_fixed 0.15 -> 0.10 ms, compute 0.19 -> 0.25 -- the sort it then waits for)


def run(points=100_000, frames=30):
    """the tracker cycle of multi_tracker_impl.cpp:83-123 with everything resident in HBM; returns the JSON object"""
    import types

    args = types.SimpleNamespace(points=points, frames=frames)
    b = pkg.scene_binding(0)
    scene, clipped, meas = mapping.Scene(b, 3), mapping.Scene(b, 3), mapping.Scene(b, 3)
    mg = mapping.MergerCorrespondenceHomo(b, mapping.MergerParams(50.0, 0.0025, 0))  # merge close points, never append
    cl = mapping.SceneClipperBall(b, range_max=50.0)
    al = pkg.MultiAligner(abi.SE3_QUAT_RIGHT, 0)
    c = abi.default_slice_config(abi.SE3_QUAT_RIGHT)
    c.kind, c.finder_max_distance, c.robustifier, c.robustifier_chi_threshold = abi.SLICE_P2PLANE, 0.25, abi.ROBUST_CAUCHY, 0.05
    si = al.add_slice(c)
    poses = [syn.se3(np.array([0.03 * k, -0.02 * k, 0.01 * k]), np.deg2rad(np.array([0.3 * k, -0.2 * k, 0.4 * k])))
             for k in range(args.frames + 1)]
    frames = []
    for k in range(args.frames + 1):
        P, N = syn.scene_3d(args.points, 900 + k)
        Xi = syn.se3_inv(poses[k])
        frames.append((np.ascontiguousarray(P @ Xi[:, :3].T + Xi[:, 3], np.float32), np.ascontiguousarray(N @ Xi[:, :3].T, np.float32)))
    meas.set(*frames[0])
    mg.set_scene(scene); mg.set_measurement(meas); mg.set_measurement_in_scene(syn.identity(3))
    mg.compute()
    est = syn.identity(3).astype(np.float32)
    t = {"upload": 0.0, "clip": 0.0, "set_moving": 0.0, "set_fixed": 0.0, "compute": 0.0, "merge": 0.0}
    for k in range(1, args.frames + 1):
        t0 = time.perf_counter(); meas.set(*frames[k]); t1 = time.perf_counter()
        cl.set_full_scene(scene); cl.set_clipped_scene_in_robot(clipped); cl.set_robot_in_local_map(est)
        cl.compute(); t2 = time.perf_counter()
        cp, cn, n = clipped.device_arrays()
        mp, mn, m = meas.device_arrays()
        if FIXED_FIRST:  # (the reference's order: setFixed, then setMoving -- multi_tracker_impl.cpp:97-98)
            tm0 = time.perf_counter()
            al.set_cloud_device("set_fixed", si, mp, 16, mn, 16, m, kept=KEPT); tm1 = time.perf_counter()
            al.set_cloud_device("set_moving", si, cp, 16, cn, 16, n, kept=KEPT); t4 = time.perf_counter()
            t3 = t2 + (t4 - tm1)  # (so that t3 - t2 is set_moving's share and t4 - t3 set_fixed's)
        else:
            al.set_cloud_device("set_moving", si, cp, 16, cn, 16, n, kept=KEPT); t3 = time.perf_counter()
            al.set_cloud_device("set_fixed", si, mp, 16, mn, 16, m, kept=KEPT); t4 = time.perf_counter()
        al.set_moving_in_fixed(syn.identity(3))
        al.compute(); t5 = time.perf_counter()
        X = np.vstack([al.moving_in_fixed(), [0, 0, 0, 1]]).astype(np.float64)
        est = (np.vstack([est, [0, 0, 0, 1]]).astype(np.float64) @ np.linalg.inv(X))[:3].astype(np.float32)
        mg.set_measurement_in_scene(est)
        mg.compute_from_aligner(al, si, clipped); t6 = time.perf_counter()
        if k > 2:  # skip warm-up frames
            for name, d in zip(t, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
                t[name] += d
    nf = args.frames - 2
    ms = {k: 1e3 * v / nf for k, v in t.items()}
    on_device = sum(v for k, v in ms.items() if k != "upload")
    err = float(np.max(np.abs(est - poses[args.frames][:3].astype(np.float32))))
    return {"points_per_frame": args.points, "scene_points": scene.size(), "ms": ms, "ms_per_frame_on_device": on_device,
            "frames_per_s_on_device": 1e3 / on_device, "pose_error_after_%d_frames" % args.frames: err, "status": al.status()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=100_000)
    ap.add_argument("--frames", type=int, default=30)
    args = ap.parse_args()
    print(json.dumps(run(args.points, args.frames)))


if __name__ == "__main__":
    main()
