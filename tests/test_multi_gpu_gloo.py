"""world_size 2, gloo: the N>1 path of the batched loop-closure alignment -- sharding k -> k mod G and the ONE exchange of
result records at the end (all-gather, or the all-reduce(sum) form of BASELINE.json's north_star) -- must reproduce the
single-process results exactly.  CPU legs shard the ORACLE backend (no GPU here); the -m gpu legs shard the PRODUCT
backend (two ranks on device 0, gloo for the collective: RCCL refuses two ranks on one GPU; the 8-GPU RCCL run is the
driver's) and compare the table's bytes with the single-process product run."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_alignments(indices, K, backend):
    from helpers import cue_config
    from srrg2_slam_interfaces_amd import _abi as abi
    from srrg2_slam_interfaces_amd import distributed as D
    from srrg2_slam_interfaces_amd import synthetic as syn

    probs = syn.batch_3d(K=K, n=1500, seed=4300)
    if backend == "oracle":
        from oracle import pyoracle

        al = pyoracle.OracleAligner(abi.SE3_QUAT_RIGHT)
    else:
        import srrg2_slam_interfaces_amd as pkg

        al = pkg.MultiAligner(abi.SE3_QUAT_RIGHT, device=0)
    si = al.add_slice(cue_config(abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE, 0.35))
    al.set_fixed(si, probs[0]["fixed"], probs[0]["fixed_normals"])
    mine = [probs[k] for k in indices]
    if not mine:
        return []
    res = al.compute_batch([p["moving"] for p in mine], [syn.identity(3)] * len(mine),
                           [p["moving_normals"] for p in mine])
    return [D.pack_record(k, r) for k, r in zip(indices, res)]


def _worker(rank, world, port, K, out_dir, backend):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from srrg2_slam_interfaces_amd import distributed as D

    recs = _run_alignments(D.shard(K, world, rank), K, backend)
    np.save(os.path.join(out_dir, "gather_%d.npy" % rank), D.all_gather_records(recs, K))
    np.save(os.path.join(out_dir, "reduce_%d.npy" % rank), D.all_reduce_records(recs, K))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_covers_every_alignment_once():
    from srrg2_slam_interfaces_amd import _capi
    from srrg2_slam_interfaces_amd import distributed as D

    lib = _capi.lib()
    for K in (0, 1, 7, 256):
        for world in (1, 2, 8):
            seen = sorted(k for r in range(world) for k in D.shard(K, world, r))
            assert seen == list(range(K))
            for r in range(world):  # the C ABI's rule is the same rule
                n = lib.srrg2_multi_gpu_shard_count(K, world, r)
                buf = (C.c_int32 * max(n, 1))()
                assert lib.srrg2_multi_gpu_shard_indices(K, world, r, buf) == n
                assert list(buf[:n]) == D.shard(K, world, r)
    assert D.shard(256, 8, 3)[:3] == [3, 11, 19] and len(D.shard(256, 8, 3)) == 32
    assert lib.srrg2_multi_gpu_shard_count(5, 0, 0) < 0


def test_record_layout_is_the_c_abis():
    """distributed.pack_record (host side) and srrg2_multi_gpu_pack_record / unpack_record (C ABI) agree"""
    from srrg2_slam_interfaces_amd import _abi as abi
    from srrg2_slam_interfaces_amd import _capi
    from srrg2_slam_interfaces_amd import distributed as D

    lib = _capi.lib()
    assert D.RECORD_FLOATS == 41
    rng = np.random.default_rng(7)
    for kind, tsize, Dd in ((abi.SE3_QUAT_RIGHT, 12, 6), (abi.SE2_RIGHT, 9, 3)):
        r = abi.BatchResult()
        for i in range(tsize):
            r.moving_in_fixed[i] = float(np.float32(rng.normal()))
        r.status, r.num_iterations, r.num_correspondences = 0, 10, 4321
        r.last.num_inliers, r.last.num_outliers, r.last.chi_inliers = 4000, 321, 0.125
        A = rng.normal(size=(Dd, Dd)).astype(np.float32)
        H = (A @ A.T).astype(np.float32)
        H = np.triu(H) + np.triu(H, 1).T
        for i, v in enumerate(H.reshape(-1)):
            r.information[i] = float(v)
        rec = (C.c_double * 41)()
        assert lib.srrg2_multi_gpu_pack_record(17, kind, C.byref(r), rec) == 0
        shape = (3, 3) if tsize == 9 else (3, 4)
        py = D.pack_record(17, {"moving_in_fixed": np.array(r.moving_in_fixed[:tsize], np.float32).reshape(shape),
                                "status": 0, "num_iterations": 10, "num_correspondences": 4321, "information": H,
                                "last": {"num_inliers": 4000, "num_outliers": 321, "num_correspondences": 9999,
                                         "chi_inliers": 0.125}})
        assert np.array_equal(np.array(rec[:]), py)
        back, k = abi.BatchResult(), C.c_int(-1)
        assert lib.srrg2_multi_gpu_unpack_record(rec, kind, C.byref(k), C.byref(back)) == 0
        assert k.value == 17 and back.num_correspondences == 4321 and back.last.num_inliers == 4000
        assert np.array_equal(np.array(back.information[:Dd * Dd], np.float32).reshape(Dd, Dd), H)
        u = D.unpack_record(py, tsize)
        assert u["k"] == 17 and np.array_equal(u["information"], H)


def _check_tables(tmp_path, K, backend):
    from srrg2_slam_interfaces_amd import distributed as D

    single = D.all_gather_records(_run_alignments(list(range(K)), K, backend), K)
    for form in ("gather", "reduce"):
        t0 = np.load(tmp_path / ("%s_0.npy" % form))
        t1 = np.load(tmp_path / ("%s_1.npy" % form))
        assert t0.tobytes() == t1.tobytes()      # every rank holds the full table
        assert t0.tobytes() == single.tobytes()  # and it equals the unsharded run bit for bit (X, statistics, H)
    t0 = np.load(tmp_path / "reduce_0.npy")
    for k in range(K):
        r = D.unpack_record(t0[k])
        assert r["k"] == k and r["status"] == 0 and r["num_iterations"] == 10
        H = r["information"]
        assert np.all(np.linalg.eigvalsh(H.astype(np.float64)) > 0)  # information of a converged alignment: SPD
    return single


def test_two_ranks_gloo_equal_single_process(tmp_path):
    K, world = 5, 2
    mp.spawn(_worker, args=(world, _free_port(), K, str(tmp_path), "oracle"), nprocs=world, join=True)
    _check_tables(tmp_path, K, "oracle")


@pytest.mark.gpu
def test_two_ranks_shard_the_product_backend(tmp_path):
    """the same check with the HIP library doing the alignments in both ranks (device 0), and against the oracle's table"""
    K, world = 6, 2
    mp.spawn(_worker, args=(world, _free_port(), K, str(tmp_path), "product"), nprocs=world, join=True)
    table = _check_tables(tmp_path, K, "product")
    from srrg2_slam_interfaces_amd import distributed as D

    ref = D.all_gather_records(_run_alignments(list(range(K)), K, "oracle"), K)
    assert table.tobytes() == ref.tobytes()


@pytest.mark.gpu
def test_bench_multi_rank_control_flow_on_one_gpu():
    """bench.py under torch.distributed.run with 2 ranks (barriers, max-over-ranks timing, the exchange of the result
    records, rank 0 prints the one JSON line): N > 1 measures BASELINE's second metric, the 256-alignment batched
    loop-closure job, strong scaling.  A one-GPU box cannot run two RCCL ranks, so the test hook SRRG2_BENCH_SHARE_GPU=1
    puts both ranks on device 0 and the process group on gloo; the 8-GPU run is the driver's."""
    import json
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SRRG2_BENCH_SHARE_GPU="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(root, "bench.py"),
                          "--gpus", "2", "--steps", "3", "--warmup", "1", "--total-alignments", "16",
                          "--batch-points", "20000"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["value"] > 0
    assert rec["config"]["alignments_total"] == 16 and rec["config"]["alignments_per_step_per_gpu"] == 8
    assert rec["alignments_per_sec"] > 0 and rec["config"]["all_success"] is True
    assert "cpu_baseline" not in rec  # rank 0 at N = 1 only


@pytest.mark.gpu
def test_bench_eight_ranks_at_the_real_shape_of_the_job(tmp_path):
    """the N = 8 control flow at the REAL shape of BASELINE's second metric -- 256 alignments, 8 ranks, 32 per
    rank, ONE exchange -- dry-run on a one-GPU box (SRRG2_BENCH_SHARE_GPU=1: every rank on device 0, gloo for the
    collective; smaller clouds than the benchmark's so that eight processes share the GPU comfortably).  Every rank
    synthesises only its shard; the exchanged table must equal, byte for byte, the table of ONE process aligning all 256
    against the same query map.  No scaling number is claimed: the 8-GPU run is the driver's."""
    import json
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SRRG2_BENCH_SHARE_GPU="1")
    table_path = str(tmp_path / "table.npy")
    npts = 6000
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"),
                          "--gpus", "8", "--steps", "2", "--warmup", "1", "--total-alignments", "256",
                          "--batch-points", str(npts), "--no-one-gpu-reference", "--dump-table", table_path],
                         capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["scaling"] == "strong" and rec["config"]["all_success"] is True
    assert rec["config"]["alignments_total"] == 256 and rec["config"]["alignments_per_step_by_rank"] == [32] * 8
    table = np.load(table_path)
    # the same job in ONE process: all 256 alignments in one compute_batch against the same query map
    import srrg2_slam_interfaces_amd as pkg
    from srrg2_slam_interfaces_amd import _abi as abi
    from srrg2_slam_interfaces_amd import distributed as D
    from srrg2_slam_interfaces_amd import synthetic as syn

    sys.path.insert(0, root)
    import bench

    probs = syn.batch_3d(K=256, n=npts, seed=4000, shared_fixed_group=1 << 30)
    al = bench.make_aligner(lambda: pkg.MultiAligner(abi.SE3_QUAT_RIGHT, device=0), abi, 10)
    al.set_fixed(0, probs[0]["fixed"], probs[0]["fixed_normals"])
    res = al.compute_batch([p["moving"] for p in probs], [syn.identity(3)] * 256, [p["moving_normals"] for p in probs])
    single = D.all_gather_records([D.pack_record(k, r) for k, r in enumerate(res)], 256)
    assert table.shape == single.shape == (256, D.RECORD_FLOATS)
    assert table.tobytes() == single.tobytes()
    # the shard rule: rank r ran k = r, r + 8, ...; a shard generated on its own equals the pick out of the full list
    shard3 = syn.batch_3d(K=256, n=npts, seed=4000, shared_fixed_group=1 << 30, only=D.shard(256, 8, 3))
    assert len(shard3) == 32 and all(np.array_equal(a["moving"], probs[k]["moving"]) for a, k in zip(shard3, D.shard(256, 8, 3)))


# ---- ONE alignment sharded by moving points (second mode; include/srrg2_slam_amd.h: set_point_shard) ---------
def _point_shard_problem(n=30_000):
    from srrg2_slam_interfaces_amd import synthetic as syn

    return syn.cloud_pair_3d(n=n, seed=4400)


def _point_shard_aligner(robust):
    import srrg2_slam_interfaces_amd as pkg
    from helpers import cue_config
    from srrg2_slam_interfaces_amd import _abi as abi

    al = pkg.MultiAligner(abi.SE3_QUAT_RIGHT, device=0)
    al.set_params(max_iterations=10, min_num_inliers=10, enable_inlier_only_runs=1 if robust else 0)
    cfg = cue_config(abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE, 0.25, abi.ROBUST_CAUCHY if robust else abi.ROBUST_NONE, 0.05, 0.8)
    return al, al.add_slice(cfg)


def _point_shard_run(al, moving, normals):
    """one alignment through the batch entry point (K = 1: it also returns H of the last Gauss-Newton iteration)"""
    from srrg2_slam_interfaces_amd import synthetic as syn

    r = al.compute_batch([moving], [syn.identity(3)], [normals])[0]
    st = al.iteration_stats()
    return {"X": np.array(r["moving_in_fixed"], np.float32), "status": r["status"],
            "stats": np.array([[s[k] for k in ("num_inliers", "num_outliers", "num_suppressed", "num_correspondences",
                                               "solver_status", "chi_inliers", "chi_outliers")] for s in st], np.float64),
            "H": np.array(r["information"], np.float32)}


def _point_shard_worker(rank, world, port, out_dir, robust, npts):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from srrg2_slam_interfaces_amd import _capi
    from srrg2_slam_interfaces_amd import distributed as D
    from srrg2_slam_interfaces_amd import synthetic as syn

    d = _point_shard_problem(npts)
    al, si = _point_shard_aligner(robust)
    al.set_fixed(si, d["fixed"], d["fixed_normals"])
    # an uneven, interleaved deal: rank 0 takes two of every three points
    n = d["moving"].shape[0]
    sel = (np.arange(n) % 3 != 2) if rank == 0 else (np.arange(n) % 3 == 2)
    al.set_point_shard(D.point_shard_reducer(_capi.lib()), n)
    out = _point_shard_run(al, d["moving"][sel], d["moving_normals"][sel])
    assert out["status"] == 0
    np.savez(os.path.join(out_dir, "shard_%d.npz" % rank), **out)
    # the mode can be switched off again: the same handle then aligns its own share only (a different result)
    al.set_point_shard(None, 0)
    np.save(os.path.join(out_dir, "alone_%d.npy" % rank), _point_shard_run(al, d["moving"][sel], d["moving_normals"][sel])["X"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("robust,npts", [(False, 30_000), (True, 30_000), (False, 240_000)])
def test_point_sharded_alignment_equals_the_one_gpu_alignment(tmp_path, robust, npts):
    """two ranks (device 0, gloo) each hold a share of the moving cloud; their partial fixed-point sums are added before
    every control step (exact integers): estimate, per-iteration statistics and H equal the alignment of the whole cloud
    on one GPU bit for bit -- with the inlier-only second run and the Cauchy kernel as well"""
    from srrg2_slam_interfaces_amd import synthetic as syn

    world = 2
    # (240 000 points: rank 0 holds 160 000 and runs with the deferred-search kernel, rank 1 holds 80 000 and runs
    # without it -- the ranks may take different kernel paths, the sums do not care)
    mp.spawn(_point_shard_worker, args=(world, _free_port(), str(tmp_path), robust, npts), nprocs=world, join=True)
    d = _point_shard_problem(npts)
    al, si = _point_shard_aligner(robust)
    al.set_fixed(si, d["fixed"], d["fixed_normals"])
    ref = _point_shard_run(al, d["moving"], d["moving_normals"])
    assert ref["status"] == 0 and ref["stats"].shape[0] == (20 if robust else 10)
    for rank in range(world):
        got = np.load(tmp_path / ("shard_%d.npz" % rank))
        assert int(got["status"]) == ref["status"]
        assert got["X"].tobytes() == ref["X"].tobytes()
        assert got["stats"].tobytes() == ref["stats"].tobytes()
        assert got["H"].tobytes() == ref["H"].tobytes()
        alone = np.load(tmp_path / ("alone_%d.npy" % rank))
        assert alone.tobytes() != ref["X"].tobytes()  # (a share alone is a different problem)


def _point_shard_rccl_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    from srrg2_slam_interfaces_amd import _capi
    from srrg2_slam_interfaces_amd import distributed as D

    d = _point_shard_problem()
    al, si = _point_shard_aligner(False)
    al.set_fixed(si, d["fixed"], d["fixed_normals"])
    al.set_point_shard(D.point_shard_reducer(_capi.lib()), d["moving"].shape[0])  # (RCCL group: the on-stream path)
    out = _point_shard_run(al, d["moving"], d["moving_normals"])
    np.savez(os.path.join(out_dir, "rccl.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_point_shard_reducer_on_the_aligners_stream_with_rccl(tmp_path):
    """the RCCL form of the reduction hook (device buffer -> torch tensor -> all_reduce under the aligner's own stream ->
    back, no host wait): a one-rank RCCL group is all a one-GPU box can run -- the sums come back unchanged, so the
    alignment must equal the plain one bit for bit; what it checks is the stream ordering of hook and kernels"""
    mp.spawn(_point_shard_rccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    d = _point_shard_problem()
    al, si = _point_shard_aligner(False)
    al.set_fixed(si, d["fixed"], d["fixed_normals"])
    ref = _point_shard_run(al, d["moving"], d["moving_normals"])
    got = np.load(tmp_path / "rccl.npz")
    assert int(got["status"]) == ref["status"] == 0
    assert got["X"].tobytes() == ref["X"].tobytes()
    assert got["stats"].tobytes() == ref["stats"].tobytes()
    assert got["H"].tobytes() == ref["H"].tobytes()
