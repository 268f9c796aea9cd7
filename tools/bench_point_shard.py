#!/usr/bin/env python
"""ONE big alignment sharded by moving points over the ranks of the job (srrg2_aligner_set_point_shard).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_point_shard.py --points 4000000

One rank per GPU over RCCL; SRRG2_BENCH_SHARE_GPU=1 puts every rank on device 0 with a gloo group (a one-GPU box: checks
the control flow, not the scaling).  Rank 0 prints one JSON line: iterations/s of the sharded alignment, the same
alignment on one GPU (rank 0 alone, after the timed region), and whether the two estimates are bit-identical."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=2_000_000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--iterations", type=int, default=10)
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    share = os.environ.get("SRRG2_BENCH_SHARE_GPU") == "1"
    dev = 0 if share else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo" if share else "nccl", rank=rank, world_size=world)
    import srrg2_slam_interfaces_amd as pkg
    from srrg2_slam_interfaces_amd import _abi as abi
    from srrg2_slam_interfaces_amd import _capi
    from srrg2_slam_interfaces_amd import distributed as D
    from srrg2_slam_interfaces_amd import synthetic as syn

    d = syn.cloud_pair_3d(n=a.points, seed=2000)
    ident = syn.identity(3)

    def aligner():
        al = pkg.MultiAligner(abi.SE3_QUAT_RIGHT, device=dev)
        al.set_params(max_iterations=a.iterations, min_num_inliers=10)
        c = abi.default_slice_config(abi.SE3_QUAT_RIGHT)
        c.kind, c.finder, c.finder_max_distance, c.finder_normal_cos = abi.SLICE_P2PLANE, abi.FINDER_NN_GATED, 0.25, 0.8
        c.robustifier, c.robustifier_chi_threshold = abi.ROBUST_CAUCHY, 0.05
        si = al.add_slice(c)
        al.set_fixed(si, d["fixed"], d["fixed_normals"])
        return al, si

    def run(al, si, steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            al.set_moving_in_fixed(ident)
            assert al.compute() == 0
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    al, si = aligner()
    sel = np.arange(a.points) % world == rank
    al.set_moving(si, d["moving"][sel], d["moving_normals"][sel])
    al.set_point_shard(D.point_shard_reducer(_capi.lib()), a.points)
    run(al, si, a.warmup)
    dist.barrier()
    dt = run(al, si, a.steps)
    t = torch.tensor([dt], dtype=torch.float64, device="cuda" if not share else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    X = np.array(al.moving_in_fixed(), np.float32)
    if rank == 0:
        al1, s1 = aligner()
        al1.set_moving(s1, d["moving"], d["moving_normals"])
        run(al1, s1, a.warmup)
        dt1 = run(al1, s1, a.steps)
        X1 = np.array(al1.moving_in_fixed(), np.float32)
        print(json.dumps({"metric": "icp_iterations_per_sec", "workload": "one %d-point alignment sharded by moving points" % a.points,
                          "n_gpus": world, "shared_gpu": share, "value": a.iterations * a.steps / float(t.item()),
                          "ms_per_compute": 1e3 * float(t.item()) / a.steps,
                          "one_gpu_value": a.iterations * a.steps / dt1, "one_gpu_ms_per_compute": 1e3 * dt1 / a.steps,
                          "speedup": dt1 / float(t.item()), "bit_identical_to_one_gpu": X.tobytes() == X1.tobytes()}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
