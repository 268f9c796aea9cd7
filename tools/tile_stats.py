#!/usr/bin/env python
"""Census of the wave tiles of k_icp_step_tile per search iteration, from a -DSRRG2_TILE_STATS build:
  make -C srrg2_slam_interfaces_amd/csrc OUT=../lib/libsrrg2_slam_amd_stats.so EXTRA=-DSRRG2_TILE_STATS
  SRRG2_AMD_LIB=srrg2_slam_interfaces_amd/lib/libsrrg2_slam_amd_stats.so SRRG2_AMD_LDS_TILE=1 python tools/tile_stats.py [K]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import srrg2_slam_interfaces_amd as pkg
from srrg2_slam_interfaces_amd import _abi as abi, _capi, synthetic as syn

K = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
lib = _capi.lib()
al = pkg.MultiAligner(abi.SE3_QUAT_RIGHT)
al.set_params(max_iterations=10, min_num_inliers=10)
c = abi.default_slice_config(abi.SE3_QUAT_RIGHT)
c.kind, c.finder, c.finder_max_distance, c.finder_normal_cos = abi.SLICE_P2PLANE, abi.FINDER_NN_GATED, 0.25, 0.8
c.robustifier, c.robustifier_chi_threshold = abi.ROBUST_CAUCHY, 0.05
al.add_slice(c)
probs = syn.batch_3d(K=K, n=n, seed=4000, shared_fixed_group=1 << 30)
al.set_fixed(0, probs[0]["fixed"], probs[0]["fixed_normals"])
buf = (C.c_uint64 * 64)()
lib.srrg2_amd_debug_tile_stats(buf, 1)
al.compute_batch([p["moving"] for p in probs], np.stack([syn.identity(3)] * K), [p["moving_normals"] for p in probs])
lib.srrg2_amd_debug_tile_stats(buf, 0)
names = ["waves need1", "ok", "fail cells/row", "fail rows", "fail candidates", "sum candidates", "sum rows", "sum lanes",
         "waves need2", "ok2", "fail2 cells/row", "fail2 rows", "fail2 candidates", "sum candidates2", "sum lanes2", "coop lanes"]
for it in range(4):
    v = [buf[it * 16 + k] for k in range(16)]
    if not v[0]:
        continue
    print("iteration %s: " % (it if it < 3 else "3+") + ", ".join("%s %d" % (nm, x) for nm, x in zip(names, v)))
    print("   phase 1: %.1f%% staged, %.0f candidates / %.1f rows / %.1f lanes per wave; shell: %.1f%% of waves, %.1f%% staged, %.0f candidates, %.1f lanes"
          % (100.0 * v[1] / v[0], v[5] / v[0], v[6] / v[0], v[7] / v[0], 100.0 * v[8] / v[0], 100.0 * v[9] / max(v[8], 1), v[13] / max(v[8], 1), v[14] / max(v[8], 1)))
v = [buf[48 + k] for k in range(4)]
if v[0]:
    print("first phase on staged tiles: %d waves; groups of four candidates executed per wave: %.1f row by row, %.1f if every lane "
          "walked its own flattened list, %.1f on average per lane" % (v[0], v[1] / v[0], v[2] / v[0], v[3] / v[0] / 64.0))
