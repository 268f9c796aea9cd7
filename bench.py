#!/usr/bin/env python
"""bench.py -- headline benchmark of the aligner hot path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...; a bare
   `python bench.py --gpus N` launches those N ranks itself.  `n_gpus` on the line is the size of the process group that
   ran, and that group must have N ranks on N distinct devices over RCCL -- anything else fails loudly, never N = 1 quietly)

BASELINE.json's metric has two halves and the default workload follows N:
  N = 1  "ICP iterations/sec (100k-pt SE(3) point-to-plane) @1 GPU": config C2.  One step = set the guess + one blocking
         MultiAligner::compute() (S/registration/aligners/multi_aligner_impl.cpp:47-95) = `iterations` ICP iterations
         (finder + linearise/reduce + 6x6 solve + update) on clouds resident in HBM + read status and estimate.
  N > 1  "8-GPU batched aligns/sec": config C4, the loop-closure candidate batch of
         multi_loop_detector_brute_force_impl.cpp:64-91 -- a FIXED job of --total-alignments (256) independent 50k-point
         alignments against one query map, sharded k -> k mod N (STRONG scaling: 32 per GPU at N = 8), one
         compute_batch() per rank per step, no collective on the data path; the result records (X, statistics, H) are
         exchanged ONCE after the timed region by an all-reduce(sum) (distributed.py).  After the timed region rank 0
         also runs the whole job alone for a few steps, so that every line carries its own one-GPU reference.
`--workload c2|c3|c4` overrides (c4 at N = 1: --batch alignments on the one GPU).

Prints ONE JSON line on rank 0.  `value` = ICP iterations/s over all ranks (`alignments_per_sec` beside it for C4).
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
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--points", type=int, default=100_000)
    ap.add_argument("--iterations", type=int, default=10)  # aligner.h:30
    ap.add_argument("--workload", default=None, choices=["c2", "c3", "c4", "c5"])
    ap.add_argument("--batch", type=int, default=32, help="c4 at N = 1: alignments per step")
    ap.add_argument("--total-alignments", type=int, default=256, help="c4 at N > 1: the job all ranks share")
    ap.add_argument("--batch-points", type=int, default=50_000)
    ap.add_argument("--cell-size", type=float, default=0.0)
    ap.add_argument("--overlap", type=float, default=1.0, help="c2 experiment: keep this x-quantile of the fixed cloud")
    ap.add_argument("--exchange", default="all_reduce", choices=["all_reduce", "all_gather"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-all-cores", action="store_true")
    ap.add_argument("--no-one-gpu-reference", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--c3-own-clouds", action="store_true", help="c3: every slice uploads its own copy of the clouds (round <= 3)")
    ap.add_argument("--dump-table", default=None, help="rank 0 saves the exchanged result table of the last step (.npy): tests")
    ap.add_argument("--print-launch", action="store_true", help="N > 1 without a launcher: print the command that would be run, and exit")
    a = ap.parse_args()
    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" in os.environ and world != a.gpus:
        # (a launcher started `world` ranks for a command line that says --gpus N: the line would claim the wrong N)
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (a.gpus, world))
    a.explicit_workload = a.workload is not None
    if a.workload is None:
        a.workload = "c2" if world == 1 else "c4"
    fill_defaults(a, world)
    return a


def fill_defaults(a, world):
    if a.steps is None:
        a.steps = 200 if a.workload == "c2" else (100 if a.workload == "c3" else 20)
    if a.warmup is None:
        a.warmup = 20 if a.workload != "c4" else 3


def launch_command(gpus, argv, port):
    """`python bench.py --gpus N ...` without a launcher: the command that starts the N ranks (one per GPU, RCCL)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(args):
    """--gpus N > 1 and no WORLD_SIZE in the environment: start the N ranks (the driver may run the multi-GPU points the way
    it runs N = 1).  Returns the exit code of the launcher; rank 0's JSON line passes through on stdout."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    argv = [x for x in sys.argv[1:] if x != "--print-launch"]
    cmd = launch_command(args.gpus, argv, port)
    if args.print_launch:
        print(json.dumps({"launch": cmd}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    return subprocess.call(cmd, env=env)


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


def measure(args, init_dist=True):
    """one workload (args.workload) on this rank's GPU: the bench line of that workload as a dict (rank 0), None elsewhere"""
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
    elif world > torch.cuda.device_count():
        raise SystemExit("bench.py: %d ranks but only %d HIP device(s) visible: one rank per GPU (a one-GPU dry run of the "
                         "control flow needs SRRG2_BENCH_SHARE_GPU=1 and says so on its line)" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dist = None
    coll_device = "cpu" if share else "cuda"
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if init_dist:
            if share:
                dist.init_process_group(backend="gloo")
            else:
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        # the line says n_gpus = N only when N ranks really ran, over RCCL unless this is the one-GPU dry run
        if dist.get_world_size() != args.gpus or world != args.gpus:
            raise SystemExit("bench.py: --gpus %d but the process group has %d ranks" % (args.gpus, dist.get_world_size()))
        if not share and dist.get_backend() != "nccl":
            raise SystemExit("bench.py: the %d-rank run must use RCCL (backend nccl), got %s" % (world, dist.get_backend()))

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
            if si == 0 or args.c3_own_clouds:
                al.set_fixed(si, data["fixed"], data["fixed_normals"])
                al.set_moving(si, data["moving"], data["moving_normals"])
            else:
                # both slices read the SAME depth cloud: in the reference they would name the same scene properties
                # (fixed_slice_name / moving_slice_name, aligner_slice_processor_base.h:41-53) and bind to the same objects
                al.share_clouds(si, 0)
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
        # C4: K_total alignments sharded k -> k mod G; this rank's moving clouds are resident in HBM.  At N > 1 the job is
        # fixed (strong scaling); every rank generates the same seeded problems and keeps its shard.
        K_total = args.total_alignments if world > 1 else args.batch
        mine = D.shard(K_total, world, rank)
        # every rank synthesises ITS shard only (problem k is seeded by k: the same clouds as in the full list); the query
        # map is shared by all alignments of the job and made by every rank (setFixed once per detector call)
        probs = dict(zip(mine, syn.batch_3d(K=K_total, n=args.batch_points, seed=4000, shared_fixed_group=1 << 30, only=mine)))
        fixed_of_job = probs[mine[0]] if mine else syn.batch_3d(K=K_total, n=args.batch_points, seed=4000, shared_fixed_group=1 << 30, only=[0])[0]
        al.set_fixed(0, fixed_of_job["fixed"], fixed_of_job["fixed_normals"])

        def resident(indices):
            missing = [k for k in indices if k not in probs]
            if missing:  # (rank 0's one-GPU reference of the whole job, after the timed region)
                probs.update(zip(missing, syn.batch_3d(K=K_total, n=args.batch_points, seed=4000, shared_fixed_group=1 << 30, only=missing)))
            c = torch.from_numpy(np.concatenate([probs[k]["moving"] for k in indices], axis=0)).cuda()
            n = torch.from_numpy(np.concatenate([probs[k]["moving_normals"] for k in indices], axis=0)).cuda()
            o = np.arange(len(indices) + 1, dtype=np.int32) * args.batch_points
            return c, n, o, np.stack([ident] * len(indices))

        coords, normals, offsets, guesses = resident(mine) if mine else (None, None, None, None)
        units_per_step = args.iterations * len(mine)
        alg_bytes_per_launch = len(mine) * 48 * args.batch_points

        def step():  # (the batch call returns status, estimate, last statistics and H of every alignment)
            if not mine:
                return []
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
    D.exchange_records(records(res), K_total, device=coll_device, mode=args.exchange)  # (warms the group up, untimed)
    # (a collection of the interpreter inside the timed loop -- the clouds of a 256-alignment batch are hundreds of numpy
    # arrays -- shows up as a multi-millisecond step: collect before, keep the collector off while the steps are timed;
    # collected BEFORE the barrier + synchronize that open the timed region, not between them and the first step)
    import gc

    gc.collect()
    gc.disable()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    step_s = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        res = step()
        step_s.append(time.perf_counter() - ts)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the driver runs --steps 20, a 5 ms sample of C2.  `value` stays the K timed steps the contract asks for; a
    # second figure from a loop of >= 50 ms (>= 200 steps) rides beside it with its own min / median / max.
    long_run = None
    if world == 1 and args.workload in ("c2", "c3") and args.steps < 200:
        n_long, long_s = 200, []
        while True:
            tl0 = time.perf_counter()
            for _ in range(n_long):
                ts = time.perf_counter()
                step()
                long_s.append(time.perf_counter() - ts)
            torch.cuda.synchronize()
            dt_long = time.perf_counter() - tl0
            if dt_long >= 0.05 or n_long >= 3200:
                break
            n_long, long_s = n_long * 2, []
        long_run = {"steps": n_long, "seconds": dt_long, "value": units_per_step * n_long / dt_long,
                    "ms_per_step": dt_long / n_long * 1e3,
                    "step_ms_min_median_max": [min(long_s) * 1e3, float(np.median(long_s)) * 1e3, max(long_s) * 1e3]}
    gc.enable()
    table = D.exchange_records(records(res), K_total, device=coll_device, mode=args.exchange)
    assert table.shape[0] == K_total
    if args.dump_table and rank == 0:
        np.save(args.dump_table, table)
    all_success = bool(np.all(table[:, 12] == 0)) and bool(np.all(table[:, 13] == args.iterations))
    total_units = args.iterations * K_total if args.workload == "c4" else units_per_step * world
    if world > 1:
        t = torch.tensor([dt], device=coll_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    status = al.status()
    stats = al.iteration_stats()
    per_rank = [len(mine)] if args.workload == "c4" else [1]
    if world > 1:  # how many alignments every rank ran per step (the shard rule made visible on the line)
        cnt = torch.zeros(world, device=coll_device, dtype=torch.int64)
        cnt[rank] = per_rank[0]
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        per_rank = [int(v) for v in cnt.cpu().tolist()]

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
        return None

    # N > 1: the same job on ONE GPU (rank 0 alone, after the timed region), so that the line carries its own reference
    # for the strong-scaling ratio
    one_gpu = None
    if world > 1 and args.workload == "c4" and not args.no_one_gpu_reference:
        c1, n1, o1, g1 = resident(list(range(K_total)))
        for _ in range(2):
            al.compute_batch_device(c1.data_ptr(), 12, n1.data_ptr(), 12, o1, g1)
        torch.cuda.synchronize()
        reps = max(2, min(5, args.steps))
        t1 = time.perf_counter()
        for _ in range(reps):
            al.compute_batch_device(c1.data_ptr(), 12, n1.data_ptr(), 12, o1, g1)
        torch.cuda.synchronize()
        one_gpu = args.iterations * K_total * reps / (time.perf_counter() - t1)

    # HBM-side bytes per launch of the timed kernels: PMC counters cannot be read from inside this process, so the
    # number comes from the committed rocprofv3 --pmc passes of this same command (tools/traffic_from_pmc.py: separate
    # FETCH_SIZE / WRITE_SIZE passes, KiB -> bytes, FETCH_SIZE doubled on gfx950 as MI355X_MICROARCH.md prescribes); only
    # reported when the passes were taken at the default problem size of the workload
    # reported when the passes were taken at this very problem size (C4: the same number of alignments per launch)
    traffic, traffic_source = None, None
    tname = "traffic_%s.json" % args.workload
    if args.workload == "c4" and len(mine) != 32:
        tname = "traffic_c4_%d.json" % len(mine)
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", tname)
    default_size = ((args.points == 100_000 and args.overlap == 1.0) if args.workload == "c2" else
                    (args.batch_points == 50_000) if args.workload == "c4" else True)
    if os.path.exists(tpath) and default_size:
        with open(tpath) as fh:
            tj = json.load(fh)
        if args.workload != "c4" or tj.get("alignments_per_launch", 32) == len(mine):
            traffic = tj["bytes_per_slice_pass"]
            traffic_source = "profiles/" + tname
    out = {
        "metric": "icp_iterations_per_sec",
        "value": total_units * args.steps / dt,
        "unit": "iterations/s",
        "n_gpus": world,
        **({"shared_gpu_dry_run": True, "collective_backend": "gloo"} if (share and world > 1) else
           ({"collective_backend": "nccl (RCCL)"} if world > 1 else {})),
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "step_ms_min_median_max": [min(step_s) * 1e3, float(np.median(step_s)) * 1e3, max(step_s) * 1e3],
        **({"value_long": long_run["value"], "long_run": long_run} if long_run else {}),
        "higher_is_better": True,
        "scaling": "strong" if (args.workload == "c4" and world > 1) else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": ("C2: SE(3) point-to-plane AlignerSlice, %d-pt synthetic cloud pair per GPU, %d ICP "
                         "iterations per compute(), gated NN 0.25 m + normal gate, Cauchy 0.05" %
                         (args.points, args.iterations)) if args.workload == "c2" else
                        ("C3: MultiAligner with 2 slices (projective + point-to-plane, projective + reprojection), "
                         "640x480 depth pair, %d iterations per compute()" % args.iterations) if args.workload == "c3" else
                        ("C4: batched loop closure, %d independent %d-pt SE(3) point-to-plane alignments against ONE query map "
                         "shared by all alignments of a launch (the brute-force detector's loop: setFixed once, "
                         "multi_loop_detector_brute_force_impl.cpp:63; SURVEY 8d's 'fixed shared per group of 8' is the "
                         "8-per-launch line c4_8), sharded k -> k mod %d (%d per GPU per step = per launch), %d iterations each" %
                         (K_total, args.batch_points, world, len(mine), args.iterations)),
            "points": args.points if args.workload == "c2" else (int(data["moving"].shape[0]) if args.workload == "c3" else args.batch_points),
            "iterations_per_step": args.iterations,
            **({"shared_clouds": not args.c3_own_clouds} if args.workload == "c3" else {}),
            "alignments_per_step_per_gpu": len(mine) if args.workload == "c4" else 1,
            "alignments_per_step_by_rank": per_rank,
            "alignments_total": K_total,
            "all_success": all_success,
            "last_status": status,
            "last_num_inliers": stats[-1]["num_inliers"] if stats else None,
            "parallelism": ("alignment k on rank k mod %d, no collective on the data path; result records (X, statistics, "
                            "H) exchanged once by %s after the timed region" % (world, args.exchange)) if world > 1 else "single GPU",
        },
        "roofline": {
            "bound": "hbm",
            "kernel": ("k_icp_step_cnl_init<3,true,4> (the first pass: compute()'s prologue inside, no k_icp_init launch) / k_icp_step_cnl<3,true,1,fused> (search passes over the cell neighbour lists) / k_icp_step_fast<3,true,1,false,fused> (converged passes): one finder+factor pass of the slice; from the second pass on the launch carries the control step of the previous iteration in its prologue (fused control steps: no control launch between passes); the last step + finalize: k_icp_final_wave" if args.workload == "c2" else
                       "k_icp_step_cnl<3,true,1,fused> (search passes over the cell neighbour lists) / k_icp_step_fast<3,true,1|2,true,fused> (converged passes): one finder+factor pass over all alignments of the launch, the control steps of the previous iteration in its first workgroups" if args.workload == "c4" else "k_proj_zbuf_fz (+ the control step of the previous iteration) + k_icp_step_proj_fused (both slices share their clouds and their association: one z-buffer pass, one step launch)"),
            "achieved": achieved,
            "peak": 8000.0,
            "unit": "GB/s",
            "frac": achieved / 8000.0,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": alg_bytes_per_launch,
            "avg_launch_ms": kern_ms,
            "launches_timed": launches.value,
            # the HIP-event brackets above hold a launch AND the gap to the next one (a launch-bound chain: C2 is ~12 dependent
            # launches of 10-25 us); rocprofv3's kernel durations of the same command, per pass kernel, from the committed summary
            **({"rocprofv3": rocprof_pass_split(args.workload)} if args.workload in ("c2", "c3") else {}),
        },
    }

    if args.workload == "c4":
        out["alignments_per_sec"] = K_total * args.steps / dt
        if one_gpu is not None:
            out["one_gpu_same_job"] = {"value": one_gpu, "unit": "iterations/s",
                                       "note": "rank 0 alone runs all %d alignments per step (after the timed region)" % K_total}
            out["strong_scaling_speedup"] = out["value"] / one_gpu
    if world == 1 and args.workload == "c2" and args.overlap == 1.0:
        # `value` is the steady state of a handle that aligns many clouds against ONE fixed cloud
        # (relocalizer, loop detector: the cell neighbour lists of the fixed cloud exist from the second compute() on).  A
        # tracker binds a NEW fixed cloud every frame (multi_tracker_impl.cpp:97-105): that compute() runs on the grid
        # kernels, with control launches.  Timed here: set_fixed (untimed: its cost is the tracker cycle's, tools/bench_tracker.py)
        # then ONE compute(), repeated.
        fresh = []
        for _ in range(30):
            al.set_fixed(0, data["fixed"], data["fixed_normals"])
            torch.cuda.synchronize()
            ts = time.perf_counter()
            step()
            fresh.append(time.perf_counter() - ts)
        fresh = fresh[5:]
        out["c2_fresh_fixed"] = {"ms_per_step": float(np.median(fresh)) * 1e3, "value": units_per_step / float(np.median(fresh)),
                                 "unit": "iterations/s",
                                 "note": "the first compute() after every set_fixed (a tracker's frame): no lists yet -> the search "
                                         "passes on the grid kernels + deferred-search kernel with a control launch each, fused "
                                         "control steps from the first converged pass on"}
        al.set_fixed(0, data["fixed"], data["fixed_normals"])
        step(); step()  # (back to the steady state for what follows)
    if world == 1 and not args.no_cpu_baseline and args.workload == "c2":
        # CPU baseline = the oracle (a port: the reference cannot be built here), single
        # thread like the reference, same clouds, same iteration count; bounded sample.
        # built for THIS host's cores (-march=native) from the committed oracle sources; the
        # prebuilt x86-64-v3 library is the fallback when there is no compiler on the box
        import glob
        import subprocess
        import tempfile

        flags = "-O3 -march=native -ffp-contract=off -fno-fast-math -fPIC -std=c99"
        try:
            native = os.path.join(tempfile.mkdtemp(prefix="oracle_native_"), "liboracle.so")
            srcs = sorted(glob.glob(os.path.join(ROOT, "oracle", "o_*.c")))
            subprocess.check_call(["gcc"] + flags.split() + ["-shared", "-o", native] + srcs + ["-lm"],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            os.environ["SRRG2_ORACLE_LIB"] = native
        except Exception:
            flags = "-O3 -march=x86-64-v3 -ffp-contract=off (prebuilt: no compiler on this box)"
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
            "sample": "%d compute() calls x %d iterations on the full %d-pt C2 pair (%.1f s), oracle/*.c built here with "
                      "gcc %s, voxel-grid finder" % (n, args.iterations, args.points, cdt, flags),
            "host_cpus": os.cpu_count(),
            "parity_indices_bit_exact_and_X_within_1e-5": bool(parity),
        }
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        if not args.no_cpu_all_cores:
            # secondary number: the same single-threaded oracle on every host CPU at once, one independent
            # alignment stream per process; run in a helper process (no HIP state is forked)
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
    if world > 1:
        dist.destroy_process_group()
    return out


def rocprof_pass_split(workload):
    """mean kernel duration of the pass kernels from the committed rocprofv3 --kernel-trace --stats summary of `bench.py --workload
    <w>` (profiles/r10/r10final_rocprofv3_<w>_summary.txt; tools/profile_round.sh) and the roofline fraction that goes with it; the
    passes from the second on carry the control step of the previous iteration (~3.2 us on one wave, profiles/r6e_pass_timeline_c2.txt)"""
    path = None
    for rel in (("r10", "r10final"), ("r9", "r9final")):  # (the newest committed end-of-round set)
        cand = os.path.join(ROOT, "profiles", rel[0], "%s_rocprofv3_%s_summary.txt" % (rel[1], workload))
        if os.path.exists(cand):
            path = cand
            break
    if path is None:
        return None
    rows = []
    for line in open(path):
        f = [x.strip() for x in line.split("|")]
        if len(f) == 5 and f[1].isdigit() and any(k in f[0] for k in ("k_icp_step_fast<", "k_icp_step_cnl<", "k_icp_step_cnl_init<", "k_icp_step_proj_fused<", "k_proj_zbuf_fz")) \
                and "k_proj_zbuf_fz_init" not in f[0]:  # (the first iteration's z-buffer pass, prologue inside: one launch of eleven)
            rows.append((f[0].replace("void ", "").split("(")[0], int(f[1]), float(f[3])))
    if not rows:
        return None
    if workload == "c2":
        calls = sum(r[1] for r in rows)
        mean_us = sum(r[1] * r[2] for r in rows) / calls  # one launch per pass
        alg = 4.8e6
    else:  # c3: a pass = z-buffer launch + step launch
        mean_us = sum(r[2] for r in rows)
        alg = None
    out = {"source": os.path.relpath(path, ROOT), "kernels_avg_us": {r[0]: r[2] for r in rows}, "mean_pass_us": mean_us,
           "note": "kernel time only (no launch gaps); passes from the second on include the previous iteration's control step"}
    if alg:
        out["frac_of_8TBs_on_kernel_time"] = alg / (mean_us * 1e-6) / 8e12
    return out


def measure_c4_distinct(iterations=10, n_align=8, points=50_000):
    """SURVEY 8d's "all-distinct" C4 variant (VERDICT r4 "missing" #5): every alignment has its OWN fixed cloud.  Both batch
    callers of the reference share the fixed scene (multi_loop_detector_brute_force_impl.cpp:63, multi_relocalizer_impl.cpp:74),
    and compute_batch mirrors that; distinct fixed clouds are the loop `setFixed; setMoving; compute` of a tracker-like caller:
    each alignment pays its grid build and runs on the grid kernels (no lists for a cloud that is aligned against once)."""
    import torch

    import srrg2_slam_interfaces_amd as pkg
    from srrg2_slam_interfaces_amd import _abi as abi
    from srrg2_slam_interfaces_amd import synthetic as syn

    probs = syn.batch_3d(K=n_align, n=points, seed=4000, shared_fixed_group=1)
    al = make_aligner(lambda: pkg.MultiAligner(abi.SE3_QUAT_RIGHT, device=0), abi, iterations)
    ident = syn.identity(3)
    dev = [tuple(torch.from_numpy(np.ascontiguousarray(p[k])).cuda() for k in ("fixed", "fixed_normals", "moving", "moving_normals"))
           for p in probs]
    torch.cuda.synchronize()

    def one_round():
        ok = True
        for f, fn, m, mn in dev:  # (clouds resident in HBM, as in the batched lines)
            al.set_cloud_device("set_fixed", 0, f.data_ptr(), 12, fn.data_ptr(), 12, f.shape[0])
            al.set_cloud_device("set_moving", 0, m.data_ptr(), 12, mn.data_ptr(), 12, m.shape[0])
            al.set_moving_in_fixed(ident)
            ok = (al.compute() == abi.SUCCESS) and ok
        return ok

    one_round()
    times = []
    ok = True
    for _ in range(5):
        t0 = time.perf_counter()
        ok = one_round() and ok
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    return {"value": iterations * n_align / dt, "unit": "iterations/s", "ms_per_alignment": dt / n_align * 1e3,
            "all_success": bool(ok),
            "config": {"workload": "C4, all-distinct variant (SURVEY 8d): %d alignments of %d points, each against its OWN fixed cloud "
                                   "(set_fixed + set_moving + compute() per alignment, clouds resident in HBM)" % (n_align, points)}}


def measure_c5(cpu_seconds=30.0, with_cpu=True):
    """C5: pose-graph Gauss-Newton solve, 50 000 SE(3) poses / 200 000 factors, 10 iterations, PCG tolerance 1e-6
    (MultiGraphSLAM_::optimize -> global_solver->compute(), S/system/multi_graph_slam_impl.cpp:300-317)"""
    import srrg2_slam_interfaces_amd as pkg
    from srrg2_slam_interfaces_amd import _abi as abi
    from srrg2_slam_interfaces_amd import posegraph as pgm
    from srrg2_slam_interfaces_amd import synthetic as syn

    V, E = 50_000, 200_000
    g = syn.pose_graph_3d(V=V, E=E, seed=5000)
    E = int(g["ij"].shape[0])
    pg = pkg.PoseGraph(abi.SE3_QUAT_RIGHT)
    pg.set_tuning(keep_structure=0)  # COLD solves: every one builds the structure of the multigrid hierarchy (matching on the host, patterns on the device)
    times, st = [], None
    for _ in range(4):
        pg.set_graph(g["poses_init"], g["ij"], g["Z"])
        t0 = time.perf_counter()
        st = pg.solve()
        times.append(time.perf_counter() - t0)
    times = times[1:]  # (the first solve also pays for first-touch allocations: warm-up; the figure is the MEDIAN of three)
    best = float(np.median(times))
    pcg = sum(s_["pcg_iterations"] for s_ in st)
    params = pgm.default_params()
    converged = all(s_["solver_status"] == 0 and s_["pcg_iterations"] < params.pcg_max_iterations and
                    s_["pcg_residual"] <= 1.01e-6 for s_ in st)
    # What MultiGraphSLAM_::optimize() does over and over (multi_graph_slam_impl.cpp:300-317): the same graph again (the
    # hierarchy's structure is kept while the topology does not change), and the graph after makeNewMap appended one
    # variable and one factor (:52-90; the structure is rebuilt: that path is not incremental yet)
    pg.set_tuning(keep_structure=1)
    warm = []
    for _ in range(3):
        pg.set_graph(g["poses_init"], g["ij"], g["Z"])
        t0 = time.perf_counter()
        pg.solve()
        warm.append(time.perf_counter() - t0)
    vid = pg.add_variable(g["poses_init"][-1])
    pg.add_factor(V - 1, vid, syn.identity(3))
    t0 = time.perf_counter()
    st_app = pg.solve()
    t_append = time.perf_counter() - t0
    # 94.8 MB per linearisation, 48.8 MB per PCG iteration (float32 spec; the solver stores float64 blocks)
    alg = 94.8e6 * len(st) + 48.8e6 * pcg
    traffic, traffic_note, dom = None, None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_c5.json")
    if os.path.exists(tpath):
        with open(tpath) as fh:
            tj = json.load(fh)
        traffic = tj["bytes_per_solve"]
        dom = {"kernel": tj["dominant_kernel"], "frac": tj["dominant_kernel_frac_of_8TBs"],
               "note": "FETCH + WRITE bytes of that kernel per launch / its average duration / 8 TB/s (PMC passes)"}
        traffic_note = "profiles/traffic_c5.json (rocprofv3 --pmc passes of tools/bench_posegraph.py): the multigrid cycle moves ~20x the bytes SURVEY 8d counts for a block-Jacobi PCG iteration"
    out = {
        "value": len(st) / best, "unit": "Gauss-Newton iterations/s", "ms_per_step": best * 1e3,
        "step_ms_min_median_max": [min(times) * 1e3, best * 1e3, max(times) * 1e3], "steps": len(times),
        "dtype": "f64", "ms_per_pcg_iteration": best * 1e3 / max(pcg, 1),
        "config": {"workload": "C5: pose-graph GN solve, %d SE(3) poses / %d binary factors, %d Gauss-Newton iterations, PCG "
                               "tolerance 1e-6 (CG preconditioned by a smoothed-aggregation multigrid V-cycle), pose 0 fixed" % (V, E, len(st)),
                   "pcg_iterations": [s_["pcg_iterations"] for s_ in st], "chi_first_last": [st[0]["chi"], st[-1]["chi"]],
                   "every_linear_solve_converged": bool(converged)},
        "roofline": {"bound": "hbm", "kernel": "one PCG iteration = k_pg_spmv + V-cycle (k_mg_op on level 0, k_mg_down2 / k_mg_up2 below it, k_mg_down2_coarsest) + "
                                               "vector kernels; "
                                               "achieved = algorithmic bytes of the whole solve / its wall time",
                     "achieved": alg / best / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": alg / best / 1e9 / 8000.0,
                     "traffic": traffic, "traffic_source": traffic_note, "algorithmic_bytes": alg, "dominant_kernel": dom},
        "warm": {"same_topology_ms": float(np.median(warm)) * 1e3,
                 "after_appending_one_variable_and_factor_ms": t_append * 1e3,
                 "pcg_iterations_after_append": [s_["pcg_iterations"] for s_ in st_app],
                 "note": "ms_per_step is the COLD solve (hierarchy structure rebuilt every time: matching on the host, sparsity patterns on the device); same_topology = "
                         "set_graph with unchanged edges keeps the structure; an append rebuilds it"},
    }
    if with_cpu:
        # CPU baseline = the oracle's solver (block-Jacobi PCG, one thread) run to CONVERGENCE on a bounded sample: the
        # first Gauss-Newton iteration of the same graph with the iteration cap lifted (the full solve is ten such)
        try:
            from oracle import pyoracle

            ref = pyoracle.OraclePoseGraph(abi.SE3_QUAT_RIGHT)
            ref.set_graph(g["poses_init"], g["ij"], g["Z"])
            p1 = pgm.default_params()
            p1.max_iterations, p1.pcg_max_iterations = 1, 100000
            t0 = time.perf_counter()
            sr = ref.solve(p1)
            cdt = time.perf_counter() - t0
            out["cpu_baseline"] = {
                "value": 1.0 / cdt, "unit": "Gauss-Newton iterations/s", "cores": 1, "kind": "port",
                "sample": "the first of the ten Gauss-Newton iterations, its linear solve run to the 1e-6 tolerance: %d "
                          "block-Jacobi PCG iterations, %.1f s (oracle/o_posegraph.c, one thread)" % (sr[0]["pcg_iterations"], cdt),
                "converged": bool(sr[0]["pcg_residual"] <= 1.01e-6), "pcg_iterations": sr[0]["pcg_iterations"],
            }
            # Like for like (VERDICT / ADVICE r3): the two sides run DIFFERENT linear solvers -- the GPU a multigrid-
            # preconditioned CG (22-51 iterations per solve), the oracle block-Jacobi CG (thousands) -- so the whole-solve
            # ratio mixes a hardware factor with an algorithmic one.  Reported apart: time per CG iteration on both sides
            # (same operator, same vectors: the hardware factor) and the iteration counts (the preconditioner's factor).
            cpu_ms_per_it = cdt * 1e3 / max(sr[0]["pcg_iterations"], 1)
            gpu_first = st[0]["pcg_iterations"]
            out["cpu_baseline"].update({
                "ms_per_pcg_iteration": cpu_ms_per_it,
                "gpu_ms_per_pcg_iteration": out["ms_per_pcg_iteration"],
                "speedup_per_pcg_iteration": cpu_ms_per_it / out["ms_per_pcg_iteration"],
                "pcg_iterations_first_gn_iteration_cpu_vs_gpu": [sr[0]["pcg_iterations"], gpu_first],
                "note": "cross-algorithm: the per-iteration ratio is the hardware factor, the iteration counts the "
                        "preconditioner's; their product is not a like-for-like speed-up and is not quoted",
                "sparse_direct_reference_point": "a sparse DIRECT solve of this same first linear system (SciPy SuperLU, minimum-degree "
                                                 "ordering, one thread: the kind of solver srrg2_solver's default is said to be) took ~19 "
                                                 "MINUTES and 4 GB for one factorisation in the build container -- 400 M non-zeros of fill on "
                                                 "this 112 x 112 x 4 lattice of closures (tests/golden/make_posegraph_golden_c5.py; the "
                                                 "full-size parity test checks against its result): the block-Jacobi PCG timed here is the "
                                                 "FASTER CPU method on this graph, not a strawman",
            })
        except Exception as e:  # (informative: never fail the bench line for it)
            out["cpu_baseline"] = {"error": repr(e)}
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    explicit = args.explicit_workload
    if args.workload == "c5":  # (replicas only: the pose graph does not shard)
        r = measure_c5(with_cpu=not args.no_cpu_baseline)
        line = {"metric": "posegraph_gauss_newton_iterations_per_sec", "n_gpus": 1, "steps": 3, "warmup": 0,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic"}
        line.update(r)
        print(json.dumps(line))
        return
    out = measure(args)
    if out is None:
        return
    if world == 1 and not explicit:
        # the default line: C2 is `value`; every other BASELINE configuration rides along as a nested object with its own
        # value / ms_per_step / roofline, measured in this same process
        import copy

        def nested(workload, **over):
            a = copy.copy(args)
            a.workload, a.no_cpu_baseline = workload, True
            a.steps, a.warmup = None, None
            for k, v in over.items():
                setattr(a, k, v)
            fill_defaults(a, world)
            r = measure(a)
            return {k: r[k] for k in ("value", "unit", "ms_per_step", "step_ms_min_median_max", "steps", "config", "roofline",
                                      "alignments_per_sec") if k in r}

        try:
            out["c3"] = nested("c3", steps=50, warmup=5)
            out["c4_256"] = nested("c4", batch=256, steps=10, warmup=2)
            out["c4_32"] = nested("c4", batch=32, steps=20, warmup=3)
            out["c4_8"] = nested("c4", batch=8, steps=20, warmup=3)
            out["c4_distinct_8"] = measure_c4_distinct(iterations=args.iterations)
            out["c5"] = measure_c5(with_cpu=not args.no_cpu_baseline)
            # the primary caller's frame (MultiTrackerBase_::align binds a NEW fixed cloud every frame, multi_tracker_impl.cpp:97-98):
            # clip the local map, set_moving / set_fixed, compute, merge -- everything resident in HBM
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_tracker

                tr = bench_tracker.run(points=100_000, frames=30)
                out["tracker_frame"] = {"tracker_frame_ms": tr["ms_per_frame_on_device"], "ms": tr["ms"], "frames_per_s": tr["frames_per_s_on_device"],
                                        "points_per_frame": tr["points_per_frame"], "status": tr["status"],
                                        "note": "host wall clock around the blocking C-ABI calls of one frame, 100 000-point measurement against the "
                                                "clipped local map; `upload` (the measurement's host-to-device copy) is not in tracker_frame_ms"}
            except Exception as e:
                out["tracker_frame"] = {"error": repr(e)}
            # a tracker's aligner as the reference configures it (S/instances.cpp:35-38): the cue slice NEXT TO a prior slice (odometry
            # prior / motion model) -- fused control steps there too since round 6 (`launches`: the same aligner with fused_control = 2,
            # i.e. one control launch per iteration as before)
            try:
                import bench_prior_cue

                pc = {}
                for name, n, three_d in (("laser_1000_beams_se2_p2p", 1000, False), ("c2_100k_se3_p2plane", 100_000, True)):
                    f = bench_prior_cue.run(n, three_d, True, reps=40)
                    l = bench_prior_cue.run(n, three_d, True, reps=40, fused_control=2)
                    c = bench_prior_cue.run(n, three_d, False, reps=40)
                    pc[name] = {"compute_ms": f["steady_ms"], "new_fixed_cloud_ms": f["new_fixed_cloud_ms"], "status": f["status"],
                                "with_control_launches_ms": [l["steady_ms"], l["new_fixed_cloud_ms"]],
                                "cue_slice_alone_ms": [c["steady_ms"], c["new_fixed_cloud_ms"]]}
                pc["note"] = "ms per compute() (10 iterations), host wall clock; first number steady state (fixed cloud kept), second on a new fixed cloud"
                out["prior_cue"] = pc
            except Exception as e:
                out["prior_cue"] = {"error": repr(e)}
            # BASELINE's second half asks for >= 6x at 8 GPUs on the 256-alignment job: at 8 GPUs every rank runs 32 per
            # launch, so the one-GPU figures bound the strong-scaling ratio from above (no collective on the data path)
            out["projected_8gpu_speedup"] = 8.0 * out["c4_32"]["value"] / out["c4_256"]["value"]
        except Exception as e:  # (the headline must survive a failure of a side configuration; it is reported, not hidden)
            out["nested_error"] = repr(e)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
