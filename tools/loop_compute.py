#!/usr/bin/env python
"""Milliseconds per compute() of the C2 pair at a given size, plain loop (median of the calls after the first three)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch  # noqa: F401  (before the library: see bench.py)
import srrg2_slam_interfaces_amd as pkg
from srrg2_slam_interfaces_amd import _abi as abi, synthetic as syn
from helpers import cue_config, setup_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
d = syn.cloud_pair_3d(n=n, seed=2000)
al = pkg.MultiAligner(abi.SE3_QUAT_RIGHT, 0)
setup_pair(al, d, cue_config(abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE, 0.25, abi.ROBUST_CAUCHY, 0.05, 0.8))
ts = []
for k in range(reps + 3):
    al.set_moving_in_fixed(syn.identity(3))
    t0 = time.perf_counter(); al.compute(); ts.append((time.perf_counter() - t0) * 1e3)
ts = ts[3:]
print("n=%d  median %.3f ms  min %.3f  max %.3f  (%.0f it/s)" % (n, np.median(ts), min(ts), max(ts), 10e3 / np.median(ts)))
