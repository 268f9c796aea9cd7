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

scan2d = "--scan2d" in sys.argv  # a 2-D scan pair, SE(2) point-to-point, a NEW fixed scan before every compute() (no lists)
args = [x for x in sys.argv[1:] if not x.startswith("--")]
n = int(args[0]) if args else (3000 if scan2d else 100000)
lib = _capi.lib()
kind = abi.SE2_RIGHT if scan2d else abi.SE3_QUAT_RIGHT
al = pkg.MultiAligner(kind)
al.set_params(max_iterations=10, min_num_inliers=10)
c = abi.default_slice_config(kind)
if scan2d:
    c.kind, c.finder, c.finder_max_distance = abi.SLICE_P2P, abi.FINDER_NN_GATED, 0.5
    pr = syn.scan_pair_2d(beams=n, sigma=0.01, seed=1234)
else:
    c.kind, c.finder, c.finder_max_distance, c.finder_normal_cos = abi.SLICE_P2PLANE, abi.FINDER_NN_GATED, 0.25, 0.8
    pr = syn.batch_3d(K=1, n=n, seed=2000, shared_fixed_group=1)[0]
c.robustifier, c.robustifier_chi_threshold = abi.ROBUST_CAUCHY, 0.05
al.add_slice(c)
al.set_fixed(0, pr["fixed"], pr.get("fixed_normals"))
al.set_moving(0, pr["moving"], pr.get("moving_normals"))
buf = (C.c_uint64 * (16 * 512 * 8))()
for rep in range(4):  # (the second compute() builds the lists; the later ones run on them)
    if scan2d:
        al.set_fixed(0, pr["fixed"], pr.get("fixed_normals"))
    al.set_moving_in_fixed(syn.identity(2 if scan2d else 3))
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
