#!/usr/bin/env python
"""Lane utilisation of the cell-neighbour-list search (cnl_search) per iteration, from a -DSRRG2_CNL_STATS build:
  make -C srrg2_slam_interfaces_amd/csrc OUT=../lib/libsrrg2_slam_amd_cnlstats.so EXTRA=-DSRRG2_CNL_STATS
  SRRG2_AMD_LIB=srrg2_slam_interfaces_amd/lib/libsrrg2_slam_amd_cnlstats.so python tools/cnl_stats.py [K] [n]
A wave runs a loop as long as its busiest lane: 'lanes' = the share of the 64 lanes with work, averaged over the wave's turns."""
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
lib.srrg2_amd_debug_cnl_stats(buf, 1)
al.compute_batch([p["moving"] for p in probs], np.stack([syn.identity(3)] * K), [p["moving_normals"] for p in probs])
lib.srrg2_amd_debug_cnl_stats(buf, 0)
for it in range(4):
    v = [int(buf[it * 16 + k]) for k in range(16)]
    if not v[0]:
        continue
    w = v[0]
    print("iteration %s: %d waves, %.1f lanes need a search, %.1f walk past the first entry" % (it if it < 3 else "3+", w, v[10] / w, v[11] / w))
    print("   first entry : %.2f turns per wave, %.1f lanes busy" % (v[1] / w, v[2] / max(v[1], 1)))
    print("   header walk : %.2f steps per wave, %.1f lanes busy (%.1f headers per walking lane)" % (v[3] / w, v[4] / max(v[3], 1), 4.0 * v[4] / max(v[11], 1)))
    print("   pool        : %.1f survivors per wave in %.2f rounds, %.2f candidate groups per round, %.1f lanes busy" % (v[6] / w, v[7] / w, v[8] / max(v[7], 1), v[9] / max(v[8], 1)))
