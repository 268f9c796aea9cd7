#!/usr/bin/env python
"""bench.py -- headline benchmark of the aligner hot path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one blocking MultiAligner::compute() (S/registration/aligners/multi_aligner_impl.cpp:47-95)
= `iterations` ICP iterations (finder + linearise/reduce + 6x6 solve + update) on clouds that are already
resident in HBM.  Workload at every N: BASELINE config C2 (SE(3) point-to-plane slice, 100k-pt synthetic
cloud pair) per rank -- rank r aligns its own seeded pair, i.e. the loop-closure
candidates of multi_loop_detector_brute_force_impl.cpp:64-91 sharded one per GPU (weak scaling, no
collective on the data path; the per-alignment result records are all-gathered after every step).

Prints ONE JSON line on rank 0.  `value` = ICP iterations/s summed over all ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--points", type=int, default=100_000)
    ap.add_argument("--iterations", type=int, default=10)  # aligner.h:30
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4"])
    ap.add_argument("--batch", type=int, default=32, help="c4: alignments per GPU per step")
    ap.add_argument("--batch-points", type=int, default=50_000)
    ap.add_argument("--cell-size", type=float, default=0.0)
    ap.add_argument("--overlap", type=float, default=1.0, help="c2 experiment: keep this x-quantile of the fixed cloud")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-all-cores", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def make_aligner(pkg_or_oracle_ctor, abi, iterations, cell_size=0.0):
    al = pkg_or_oracle_ctor()
    al.set_params(max_iterations=iterations, min_num_inliers=10)
    c = abi.default_slice_config(abi.SE3_QUAT_RIGHT)
    c.kind = abi.SLICE_P2PLANE
    c.finder = abi.FINDER_NN_GATED
    c.finder_max_distance = 0.25
    c.finder_normal_cos = 0.8
    c.finder_cell_size = cell_size
    c.robustifier = abi.ROBUST_CAUCHY
    c.robustifier_chi_threshold = 0.05
    al.add_slice(c)
    return al


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # SRRG2_BENCH_SHARE_GPU=1 (test hook: exercise the multi-rank control flow on a one-GPU box): every rank uses
    # device 0 and the process group runs on gloo with host tensors -- RCCL refuses two ranks on one GPU
    share = os.environ.get("SRRG2_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    coll_device = "cpu" if share else "cuda"
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import srrg2_slam_interfaces_amd as pkg
    from srrg2_slam_interfaces_amd import _abi as abi
    from srrg2_slam_interfaces_amd import _capi
    from srrg2_slam_interfaces_amd import synthetic as syn

    ident = syn.identity(3)
    al = make_aligner(lambda: pkg.MultiAligner(abi.SE3_QUAT_RIGHT, device=local_rank), abi, args.iterations, args.cell_size)

    from srrg2_slam_interfaces_amd import distributed as D

    if args.workload == "c2":
        data = syn.cloud_pair_3d(n=args.points, seed=2000 + 10 * rank)
        if args.overlap < 1.0:  # (experiment, not the benchmark configuration) drop the fixed points beyond an x quantile:
            import numpy as _np  # a share of the moving cloud then has no neighbour within the gate
            keep = data["fixed"][:, 0] <= _np.quantile(data["fixed"][:, 0], args.overlap)
            data["fixed"], data["fixed_normals"] = data["fixed"][keep], data["fixed_normals"][keep]
        al.set_fixed(0, data["fixed"], data["fixed_normals"])
        al.set_moving(0, data["moving"], data["moving_normals"])
        K_total = world  # one alignment per rank: alignment k lives on rank k (k mod G)
        units_per_step = args.iterations  # ICP iterations per rank per step
        alg_bytes_per_launch = 12 * args.points + 24 * args.points + 12 * args.points

        def step():  # what a tracker does per frame: set the guess, align, read status and estimate
            al.set_moving_in_fixed(ident)
            return al.compute(), al.moving_in_fixed()

        def records(res):  # (once, after the timed region)
            nst, last = al.last_iteration_stats()
            return [D.pack_record(rank, {"moving_in_fixed": res[1], "status": res[0], "num_iterations": nst, "last": last})]
    elif args.workload == "c3":
        # C3: 2-slice MultiAligner (projective + point-to-plane, projective + reprojection) on a 640x480 depth pair
        data = syn.rgbd_pair(seed=3000 + 10 * rank)
        al.clear_slices()
        for sk in (abi.SLICE_P2PLANE, abi.SLICE_REPROJECTION):
            c = abi.default_slice_config(abi.SE3_QUAT_RIGHT)
            c.kind, c.finder, c.finder_max_distance = sk, abi.FINDER_PROJECTIVE, 0.05
            for i, v in enumerate(data["K"].reshape(-1)):
                c.camera_matrix[i] = v
            c.image_rows, c.image_cols = data["rows"], data["cols"]
            c.depth_min, c.depth_max = data["depth_min"], data["depth_max"]
            si = al.add_slice(c)
            al.set_fixed(si, data["fixed"], data["fixed_normals"])
            al.set_moving(si, data["moving"], data["moving_normals"])
        K_total = world
        units_per_step = args.iterations
        nm, nf = data["moving"].shape[0], data["fixed"].shape[0]
        alg_bytes_per_launch = 12 * nm + 24 * nf + 2 * 12 * nm  # both slices in one launch pair (clouds once, C1 + C2)

        def step():  # what a tracker does per frame: set the guess, align, read status and estimate
            al.set_moving_in_fixed(ident)
            return al.compute(), al.moving_in_fixed()

        def records(res):  # (once, after the timed region)
            nst, last = al.last_iteration_stats()
            return [D.pack_record(rank, {"moving_in_fixed": res[1], "status": res[0], "num_iterations": nst, "last": last})]
    else:
        # C4: K_total alignments sharded k -> k mod G; this rank's moving clouds are resident in HBM
        K_total = args.batch * world
        mine = D.shard(K_total, world, rank)
        probs = syn.batch_3d(K=args.batch, n=args.batch_points, seed=4000 + 1000 * rank, shared_fixed_group=1 << 30)
        al.set_fixed(0, probs[0]["fixed"], probs[0]["fixed_normals"])
        coords = torch.from_numpy(np.concatenate([p["moving"] for p in probs], axis=0)).cuda()
        normals = torch.from_numpy(np.concatenate([p["moving_normals"] for p in probs], axis=0)).cuda()
        offsets = np.arange(args.batch + 1, dtype=np.int32) * args.batch_points
        guesses = np.stack([ident] * args.batch)
        units_per_step = args.iterations * args.batch
        alg_bytes_per_launch = args.batch * 48 * args.batch_points

        def step():  # (the batch call returns status, estimate and last statistics of every alignment)
            return al.compute_batch_device(coords.data_ptr(), 12, normals.data_ptr(), 12, offsets, guesses)

        def records(res):
            return [D.pack_record(k, r) for k, r in zip(mine, res)]

    # Alignments are independent: a step is the hot path over this rank's shard and nothing else.  The results stay on
    # their rank while the job runs; ONE all-gather of the result records after the timed region puts the
    # table of the last step on every rank (a per-step gather would add a latency-bound collective that the path does not
    # have: the reference's detectors consume their alignments where they were computed).
    res = step()
    for _ in range(args.warmup):
        res = step()
    D.all_gather_records(records(res), K_total, device=coll_device)  # (warms the process group up, untimed)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    table = D.all_gather_records(records(res), K_total, device=coll_device)
    assert table.shape[0] == K_total
    if world > 1:
        t = torch.tensor([dt], device=coll_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    status = al.status()
    stats = al.iteration_stats()

    # roofline of the dominant kernel (k_icp_step): HIP events on the launch stream around every launch
    lib = _capi.lib()
    import ctypes as C

    lib.srrg2_aligner_profile_enable(al._h, 1)
    for _ in range(max(3, min(10, args.steps))):
        step()
    ms, launches = C.c_double(0), C.c_int64(0)
    lib.srrg2_aligner_profile_get(al._h, C.byref(ms), C.byref(launches), 1)
    lib.srrg2_aligner_profile_enable(al._h, 0)
    kern_ms = ms.value / max(launches.value, 1)
    achieved = alg_bytes_per_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # HBM-side bytes per launch of the timed kernels: PMC counters cannot be read from inside this process, so the
    # number comes from the committed rocprofv3 --pmc passes of this same command (tools/traffic_from_pmc.py: separate
    # FETCH_SIZE / WRITE_SIZE passes, KiB -> bytes, FETCH_SIZE doubled on gfx950 as MI355X_MICROARCH.md prescribes); only
    # reported when the passes were taken at the default problem size of the workload
    traffic, traffic_source = None, None
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic_%s.json" % args.workload)
    default_size = (args.points == 100_000 and args.overlap == 1.0) if args.workload == "c2" else True
    if os.path.exists(tpath) and default_size:
        with open(tpath) as fh:
            traffic = json.load(fh)["bytes_per_slice_pass"]
        traffic_source = "profiles/traffic_%s.json" % args.workload
    out = {
        "metric": "icp_iterations_per_sec",
        "value": units_per_step * args.steps * world / dt,
        "unit": "iterations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": ("C2: SE(3) point-to-plane AlignerSlice, %d-pt synthetic cloud pair per GPU, %d ICP "
                         "iterations per compute(), gated NN 0.25 m + normal gate, Cauchy 0.05" %
                         (args.points, args.iterations)) if args.workload == "c2" else
                        ("C3: MultiAligner with 2 slices (projective + point-to-plane, projective + reprojection), "
                         "640x480 depth pair, %d iterations per compute()" % args.iterations) if args.workload == "c3" else
                        ("C4-shard: %d x %d-pt SE(3) point-to-plane alignments per GPU per step, %d iterations" %
                         (args.batch, args.batch_points, args.iterations)),
            "points": args.points if args.workload == "c2" else (int(data["moving"].shape[0]) if args.workload == "c3" else args.batch_points),
            "iterations_per_step": args.iterations,
            "alignments_per_step_per_gpu": args.batch if args.workload == "c4" else 1,
            "last_status": status,
            "last_num_inliers": stats[-1]["num_inliers"] if stats else None,
            "parallelism": "1 alignment stream per GPU, results all-gathered" if world > 1 else "single GPU",
        },
        "roofline": {
            "bound": "hbm",
            "kernel": ("k_icp_step<3,true> + k_icp_step_queue<3,true> (one finder+factor pass of the slice)" if args.workload == "c2" else
                       "k_icp_step<3,true>" if args.workload == "c4" else "k_proj_zbuf_pack + k_icp_step_proj_pack (both slices of the aligner in one launch pair)"),
            "achieved": achieved,
            "peak": 8000.0,
            "unit": "GB/s",
            "frac": achieved / 8000.0,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": alg_bytes_per_launch,
            "avg_launch_ms": kern_ms,
            "launches_timed": launches.value,
        },
    }

    if world == 1 and not args.no_cpu_baseline and args.workload == "c2":
        # CPU baseline = the oracle (a port: the reference cannot be built here), single
        # thread like the reference, same clouds, same iteration count; bounded sample.
        from oracle import pyoracle

        ref = make_aligner(lambda: pyoracle.OracleAligner(abi.SE3_QUAT_RIGHT), abi, args.iterations)
        ref.set_fixed(0, data["fixed"], data["fixed_normals"])
        ref.set_moving(0, data["moving"], data["moving_normals"])
        ref.set_moving_in_fixed(ident)
        ref.compute()  # warm-up (builds the search grid)
        # parity of this very workload before the CPU time is reported
        al.set_moving_in_fixed(ident)
        al.compute()
        cr, cg = ref.correspondences(0), al.correspondences(0)
        parity = (np.array_equal(cr["fixed_idx"], cg["fixed_idx"]) and np.array_equal(cr["moving_idx"], cg["moving_idx"])
                  and float(np.max(np.abs(ref.moving_in_fixed() - al.moving_in_fixed()))) <= 1e-5)
        n = 0
        t0 = time.perf_counter()
        while True:
            ref.set_moving_in_fixed(ident)
            ref.compute()
            n += 1
            if time.perf_counter() - t0 > args.cpu_seconds or n >= 200:
                break
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "value": n * args.iterations / cdt,
            "unit": "iterations/s",
            "cores": 1,
            "kind": "port",
            "sample": "%d compute() calls x %d iterations on the full %d-pt C2 pair (%.1f s), oracle/liboracle.so "
                      "-O3 -march=x86-64-v3 -ffp-contract=off, voxel-grid finder" % (n, args.iterations, args.points, cdt),
            "host_cpus": os.cpu_count(),
            "parity_indices_bit_exact_and_X_within_1e-5": bool(parity),
        }
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        if not args.no_cpu_all_cores:
            # secondary number: the same single-threaded oracle on every host CPU at once, one independent
            # alignment stream per process; run in a helper process (no HIP state is forked)
            import subprocess

            try:
                helper = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "cpu_all_cores.py")
                r = subprocess.run([sys.executable, helper, "--points", str(args.points), "--iterations", str(args.iterations),
                                    "--seconds", "6"], capture_output=True, text=True, timeout=180)
                allc = json.loads(r.stdout.strip().splitlines()[-1])
                out["cpu_baseline"]["all_cores"] = allc
                out["cpu_baseline"]["cpu_model"] = allc.get("cpu_model", "")
                out["speedup_vs_cpu_all_cores"] = out["value"] / allc["value"]
            except Exception as e:  # the all-core figure is informative: never fail the bench line for it
                out["cpu_baseline"]["all_cores"] = {"error": repr(e)}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
