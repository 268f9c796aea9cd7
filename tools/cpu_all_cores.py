#!/usr/bin/env python
"""All-core CPU figure for the C2 workload (SURVEY.md 8d "secondary number"): the single-threaded oracle run by one
process per host CPU on independent copies of the alignment (the parallelism the reference's callers have: independent
alignments).  Test infrastructure: called by bench.py's cpu_baseline leg only.  Prints one JSON line."""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(args):
    points, iterations, seconds, start_at = args
    import numpy as np  # noqa: F401

    from oracle import pyoracle
    from srrg2_slam_interfaces_amd import _abi as abi
    from srrg2_slam_interfaces_amd import synthetic as syn

    d = syn.cloud_pair_3d(n=points, seed=2000)
    al = pyoracle.OracleAligner(abi.SE3_QUAT_RIGHT)
    al.set_params(max_iterations=iterations)
    c = abi.default_slice_config(abi.SE3_QUAT_RIGHT)
    c.kind, c.finder_max_distance, c.robustifier, c.robustifier_chi_threshold = abi.SLICE_P2PLANE, 0.25, abi.ROBUST_CAUCHY, 0.05
    c.finder_normal_cos = 0.8
    al.add_slice(c)
    al.set_fixed(0, d["fixed"], d["fixed_normals"])
    al.set_moving(0, d["moving"], d["moving_normals"])
    al.set_moving_in_fixed(syn.identity(3))
    al.compute()  # warm-up: builds the grid
    while time.time() < start_at:  # all workers measure the same window
        time.sleep(0.01)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        al.set_moving_in_fixed(syn.identity(3))
        al.compute()
        n += 1
    return n * iterations / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=100_000)
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--procs", type=int, default=0)
    a = ap.parse_args()
    procs = a.procs or os.cpu_count() or 1
    start_at = time.time() + 4.0 + 0.01 * procs  # time for every worker to generate its clouds and warm up
    with mp.get_context("fork").Pool(procs) as pool:
        rates = pool.map(worker, [(a.points, a.iterations, a.seconds, start_at)] * procs)
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    print(json.dumps({"value": sum(rates), "unit": "iterations/s", "cores": procs, "cpu_model": model,
                      "min_per_proc": min(rates), "max_per_proc": max(rates)}))


if __name__ == "__main__":
    main()
