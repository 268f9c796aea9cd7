#!/usr/bin/env python
"""C5 measurement: pose-graph GN on the GPU vs the CPU oracle (run on the GPU box).
usage: python tools/bench_posegraph.py [V] [E] [--cpu]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import srrg2_slam_interfaces_amd as pkg
from srrg2_slam_interfaces_amd import _abi as abi
from srrg2_slam_interfaces_amd import posegraph as pgm
from srrg2_slam_interfaces_amd import synthetic as syn

_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
V = int(_pos[0]) if len(_pos) > 0 else 50000
E = int(_pos[1]) if len(_pos) > 1 else 200000
t0 = time.time()
g = syn.pose_graph_3d(V=V, E=E, seed=5000)
gen_s = time.time() - t0
pg = pkg.PoseGraph(abi.SE3_QUAT_RIGHT)
out = {"V": V, "E": int(g["ij"].shape[0]), "generate_s": gen_s}
for rep in range(3):
    pg.set_graph(g["poses_init"], g["ij"], g["Z"])
    t0 = time.perf_counter()
    st = pg.solve()
    dt = time.perf_counter() - t0
out["gpu_solve_s"] = dt
out["gpu_gn_it_per_s"] = len(st) / dt
out["gpu_pcg_its"] = [s["pcg_iterations"] for s in st]
out["gpu_chi"] = [s["chi"] for s in st]
out["gpu_pcg_it_per_s"] = sum(out["gpu_pcg_its"]) / dt
if "--cpu" in sys.argv:
    from oracle import pyoracle

    ref = pyoracle.OraclePoseGraph(abi.SE3_QUAT_RIGHT)
    ref.set_graph(g["poses_init"], g["ij"], g["Z"])
    t0 = time.perf_counter()
    sr = ref.solve()
    cdt = time.perf_counter() - t0
    out["cpu_solve_s"] = cdt
    out["cpu_pcg_its"] = [s["pcg_iterations"] for s in sr]
    out["cpu_chi"] = [s["chi"] for s in sr]
    out["speedup"] = cdt / dt
    out["max_pose_diff"] = float(np.max(np.abs(ref.poses() - pg.poses())))
print(json.dumps(out))
