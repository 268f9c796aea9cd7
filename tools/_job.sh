#!/bin/bash
cd /root/repo
python - <<'PY'
import sys, time, numpy as np
sys.path.insert(0, "tests")
import torch
import srrg2_slam_interfaces_amd as pkg
from srrg2_slam_interfaces_amd import _abi as abi, synthetic as syn
from helpers import cue_config, setup_pair
d = syn.cloud_pair_3d(n=1_000_000, seed=2000)
al = pkg.MultiAligner(abi.SE3_QUAT_RIGHT, 0)
setup_pair(al, d, cue_config(abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE, 0.25, abi.ROBUST_CAUCHY, 0.05, 0.8))
ts = []
for k in range(25):
    al.set_moving_in_fixed(syn.identity(3))
    t0 = time.perf_counter(); al.compute(); ts.append((time.perf_counter() - t0) * 1e3)
print("ms per compute:", [round(t, 2) for t in ts])
PY
