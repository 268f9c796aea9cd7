#!/usr/bin/env python
"""Where the time of a fused pass launch of ONE alignment goes (C2), from a -DSRRG2_PASS_TIMELINE build:
  make -C srrg2_slam_interfaces_amd/csrc OUT=../lib/libsrrg2_slam_amd_timeline.so EXTRA=-DSRRG2_PASS_TIMELINE
  SRRG2_AMD_LIB=srrg2_slam_interfaces_amd/lib/libsrrg2_slam_amd_timeline.so python tools/pass_timeline.py [n]
Per launch epoch: first / last workgroup start, control step done, record seen (first / last workgroup), first / last end,
in microseconds after the end of the previous launch (100 MHz constant-rate clock: 10 ns steps)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import srrg2_slam_interfaces_amd as pkg
from srrg2_slam_interfaces_amd import _abi as abi, _capi, synthetic as syn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
lib = _capi.lib()
al = pkg.MultiAligner(abi.SE3_QUAT_RIGHT)
al.set_params(max_iterations=10, min_num_inliers=10)
c = abi.default_slice_config(abi.SE3_QUAT_RIGHT)
c.kind, c.finder, c.finder_max_distance, c.finder_normal_cos = abi.SLICE_P2PLANE, abi.FINDER_NN_GATED, 0.25, 0.8
c.robustifier, c.robustifier_chi_threshold = abi.ROBUST_CAUCHY, 0.05
al.add_slice(c)
pr = syn.batch_3d(K=1, n=n, seed=2000, shared_fixed_group=1)[0]
al.set_fixed(0, pr["fixed"], pr["fixed_normals"])
al.set_moving(0, pr["moving"], pr["moving_normals"])
buf = (C.c_uint64 * (16 * 512 * 8))()
for rep in range(4):  # (the second compute() builds the lists; the later ones run on them)
    al.set_moving_in_fixed(syn.identity(3))
    lib.srrg2_amd_debug_pass_timeline(buf, 1)
    al.compute()
lib.srrg2_amd_debug_pass_timeline(buf, 0)
ts = np.frombuffer(buf, dtype=np.uint64).reshape(16, 512, 8).astype(np.int64)
prev_end = None
print("epoch | workgroups | start: first  last | control done | record seen: first  median  last | end: first  median  last | launch   [us after the end of the previous launch]")
for e in range(16):
    used = ts[e, :, 0] > 0
    used[500] = False
    census = ts[e, 500, :7].copy()
    if not used.any():
        continue
    st, ctl, seen, end = ts[e, used, 0], ts[e, 0, 1], ts[e, used, 2], ts[e, used, 3]
    seen, end = seen[seen > 0], end[end > 0]
    if not len(end):
        continue
    t0 = prev_end if prev_end is not None else st.min()
    f = lambda x: "%6.2f" % ((x - t0) / 100.0)
    print("%5d | %10d | %s  %s | %s | %s  %s  %s | %s  %s  %s | %6.2f" % (
        e, used.sum(), f(st.min()), f(st.max()), f(ctl) if ctl else "     -", f(seen.min()), f(np.median(seen)), f(seen.max()),
        f(end.min()), f(np.median(end)), f(end.max()), (end.max() - st.min()) / 100.0))
    if ts[e, 0, 5] and ctl:
        print("      control step: sums and H / b in the lanes at %s, dx solved at %s, X and the transforms at %s, published at %s" % (f(ts[e, 0, 5]), f(ts[e, 0, 6]), f(ts[e, 0, 7]), f(ctl)))
    if census[0]:
        print("      failed certificates: %d points (no previous neighbour %d; margin < 1e-4 cells %d, < 0.0202 cells %d, larger %d; moved > 1e-3 cells %d; no radius %d)" % tuple(census))
    p2 = ts[e, used, 4]
    p2 = p2[p2 > 0]
    if len(p2):
        print("      second phase (failed certificates) in %d workgroups, last one done at %s" % (len(p2), f(p2.max())))
        prev_end = max(end.max(), p2.max())
    else:
        prev_end = end.max()
