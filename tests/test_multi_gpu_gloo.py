"""CPU, world_size 2, gloo: the N>1 path of the batched loop-closure alignment -- sharding k -> k mod G and the
single all-gather of result records -- must reproduce the single-process results exactly.  The per-rank compute
uses the oracle backend here (no GPU); bench.py runs the same module with the HIP backend over RCCL."""
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


def _run_alignments(indices, K):
    from helpers import cue_config
    from oracle import pyoracle
    from srrg2_slam_interfaces_amd import _abi as abi
    from srrg2_slam_interfaces_amd import distributed as D
    from srrg2_slam_interfaces_amd import synthetic as syn

    probs = syn.batch_3d(K=K, n=1500, seed=4300)
    al = pyoracle.OracleAligner(abi.SE3_QUAT_RIGHT)
    si = al.add_slice(cue_config(abi.SE3_QUAT_RIGHT, abi.SLICE_P2PLANE, 0.35))
    al.set_fixed(si, probs[0]["fixed"], probs[0]["fixed_normals"])
    mine = [probs[k] for k in indices]
    res = al.compute_batch([p["moving"] for p in mine], [syn.identity(3)] * len(mine),
                           [p["moving_normals"] for p in mine])
    return [D.pack_record(k, r) for k, r in zip(indices, res)]


def _worker(rank, world, port, K, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from srrg2_slam_interfaces_amd import distributed as D

    recs = _run_alignments(D.shard(K, world, rank), K)
    table = D.all_gather_records(recs, K)
    np.save(os.path.join(out_dir, "table_%d.npy" % rank), table)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_covers_every_alignment_once():
    from srrg2_slam_interfaces_amd import distributed as D

    for K in (0, 1, 7, 256):
        for world in (1, 2, 8):
            seen = sorted(k for r in range(world) for k in D.shard(K, world, r))
            assert seen == list(range(K))
    assert D.shard(256, 8, 3)[:3] == [3, 11, 19] and len(D.shard(256, 8, 3)) == 32


def test_two_ranks_gloo_equal_single_process(tmp_path):
    K, world = 5, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, K, str(tmp_path)), nprocs=world, join=True)
    from srrg2_slam_interfaces_amd import distributed as D

    single = D.all_gather_records(_run_alignments(list(range(K)), K), K)
    t0 = np.load(tmp_path / "table_0.npy")
    t1 = np.load(tmp_path / "table_1.npy")
    assert np.array_equal(t0, t1)          # every rank holds the full table
    assert np.array_equal(t0, single)      # and it equals the unsharded run bit for bit
    for k in range(K):
        r = D.unpack_record(t0[k])
        assert r["k"] == k and r["status"] == 0 and r["num_iterations"] == 10


@pytest.mark.gpu
def test_bench_multi_rank_control_flow_on_one_gpu():
    """bench.py under torch.distributed.run with 2 ranks (barriers, max-over-ranks timing, the all-gather of the result
    records, rank 0 prints the one JSON line).  A one-GPU box cannot run two RCCL ranks, so the test hook
    SRRG2_BENCH_SHARE_GPU=1 puts both ranks on device 0 and the process group on gloo; the 8-GPU run is the driver's."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SRRG2_BENCH_SHARE_GPU="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(root, "bench.py"),
                          "--gpus", "2", "--steps", "5", "--warmup", "1", "--points", "20000"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0 and rec["config"]["last_status"] == 0
    assert "cpu_baseline" not in rec  # rank 0 at N = 1 only
