import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import srrg2_slam_interfaces_amd as pkg
from srrg2_slam_interfaces_amd import _abi as abi, synthetic as syn
d = syn.cloud_pair_3d(n=100000, seed=77)
al = pkg.MultiAligner(abi.SE3_QUAT_RIGHT)
c = abi.default_slice_config(abi.SE3_QUAT_RIGHT)
c.kind, c.finder, c.finder_max_distance = abi.SLICE_P2PLANE, abi.FINDER_NN_GATED, 0.25
si = al.add_slice(c)
al.set_moving(si, d["moving"], d["moving_normals"])
ts = []
for k in range(600):
    t0 = time.perf_counter(); al.set_fixed(si, d["fixed"], d["fixed_normals"]); ts.append(time.perf_counter() - t0)
ts = np.array(ts[20:]) * 1e3
print("set_fixed (host clouds) ms: median %.4f p90 %.4f p99 %.4f max %.4f; > 2x median: %d of %d" % (np.median(ts), np.percentile(ts, 90), np.percentile(ts, 99), ts.max(), int((ts > 2 * np.median(ts)).sum()), len(ts)))
