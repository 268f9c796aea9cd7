#!/usr/bin/env python
"""Batched locked-correspondence solves (MultiLoopDetectorHBST_::_computeAlignments): K candidates x M descriptor matches,
max_iterations Gauss-Newton steps each, one compute_batch_correspondences() call.  One JSON line; --cpu times the oracle
on a few candidates."""
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
from test_given_correspondences import _cfg, _landmarks  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--candidates", type=int, default=256)
    ap.add_argument("--matches", type=int, default=3000)
    ap.add_argument("--iterations", type=int, default=15)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--cpu", action="store_true")
    args = ap.parse_args()
    base = _landmarks(1, n=args.matches, outlier_ratio=0.2)
    cands = [_landmarks(100 + k, n=args.matches, outlier_ratio=0.2) for k in range(min(args.candidates, 16))]
    for c in cands:  # every reference sees the query's landmarks: the reference cloud is the fixed cloud moved by its pose
        Xi = syn.se3_inv(c["X_gt"])
        c["moving"] = (base["fixed"].astype(np.float64) @ Xi[:, :3].T + Xi[:, 3]).astype(np.float32)
    movs = [cands[k % len(cands)]["moving"] for k in range(args.candidates)]
    corrs = [cands[k % len(cands)]["corr"] for k in range(args.candidates)]
    guesses = [syn.identity(3)] * args.candidates

    def run(al, K):
        al.set_params(max_iterations=args.iterations)
        al.add_slice(_cfg(abi.SE3_QUAT_RIGHT, abi.SLICE_P2P))
        al.set_fixed(0, base["fixed"], None)
        al.compute_batch_correspondences(movs[:K], corrs[:K], guesses[:K])
        t0 = time.perf_counter()
        for _ in range(args.reps if K == args.candidates else 1):
            res = al.compute_batch_correspondences(movs[:K], corrs[:K], guesses[:K])
        return (time.perf_counter() - t0) / (args.reps if K == args.candidates else 1), res

    dt, res = run(pkg.MultiAligner(abi.SE3_QUAT_RIGHT, 0), args.candidates)
    out = {"candidates": args.candidates, "matches_each": len(corrs[0]), "iterations": args.iterations,
           "ms_per_batch_incl_upload": 1e3 * dt, "solves_per_s": args.candidates / dt,
           "gn_iterations_per_s": args.candidates * args.iterations / dt, "all_success": all(r["status"] == 0 for r in res)}
    if args.cpu:
        from oracle import pyoracle

        kc = min(8, args.candidates)
        dtc, _ = run(pyoracle.OracleAligner(abi.SE3_QUAT_RIGHT), kc)
        out["cpu_oracle_solves_per_s_1core"] = kc / dtc
        out["speedup"] = out["solves_per_s"] / out["cpu_oracle_solves_per_s_1core"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
