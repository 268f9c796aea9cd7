#!/usr/bin/env python
"""Latency of small alignments (the reference's laser-scan regime, C1 of BASELINE.json): milliseconds per compute() on the
GPU against the single-thread CPU oracle, same inputs, 10 iterations.  One JSON line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402

import srrg2_slam_interfaces_amd as pkg  # noqa: E402
from srrg2_slam_interfaces_amd import _abi as abi  # noqa: E402
from srrg2_slam_interfaces_amd import synthetic as syn  # noqa: E402
from helpers import cue_config, setup_pair  # noqa: E402


def timed(al, n, guess):
    t = 0.0
    for k in range(n + 3):
        al.set_moving_in_fixed(guess)  # every compute() starts from the same (misaligned) guess
        t0 = time.perf_counter()
        al.compute()
        if k >= 3:
            t += time.perf_counter() - t0
    return t / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--beams", type=int, nargs="*", default=[])
    ap.add_argument("--points", type=int, nargs="*", default=[2000, 10000, 30000])
    args = ap.parse_args()
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pyoracle
    cases = [("C1 2D scan 1000 beams, SE(2) p2p", abi.SE2_RIGHT, syn.scan_pair_2d(), cue_config(abi.SE2_RIGHT, abi.SLICE_P2P, 0.5))]
    for beams in args.beams:
        cases.append(("2D scan %d beams, SE(2) p2p" % beams, abi.SE2_RIGHT, syn.scan_pair_2d(beams=beams),
                      cue_config(abi.SE2_RIGHT, abi.SLICE_P2P, 0.5)))
    for n in args.points:
        cases.append(("3D %d pts, SE(3) point-to-plane" % n, abi.SE3_QUAT_RIGHT, syn.cloud_pair_3d(n=n, seed=2000),
                      cue_config(abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE, 0.25, abi.ROBUST_CAUCHY, 0.05, 0.8)))
    out = []
    for name, kind, d, cfg in cases:
        g, o = pkg.MultiAligner(kind, 0), pyoracle.OracleAligner(kind)
        setup_pair(g, d, cfg)
        setup_pair(o, d, cfg)
        guess = syn.identity(2 if kind == abi.SE2_RIGHT else 3)
        ms_g = timed(g, args.reps, guess)
        ms_o = timed(o, max(5, args.reps // 20), guess)
        same = g.moving_in_fixed().tobytes() == o.moving_in_fixed().tobytes()
        out.append({"case": name, "gpu_ms": round(ms_g, 4), "oracle_ms": round(ms_o, 4), "speedup": round(ms_o / ms_g, 2),
                    "X_bit_identical": same})
    print(json.dumps({"small_alignments": out}))


if __name__ == "__main__":
    main()
