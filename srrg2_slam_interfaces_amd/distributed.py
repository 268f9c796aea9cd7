"""Multi-GPU sharding of independent alignments.

Loop-closure candidate alignments are mutually independent (the aligner state is fully reset per compute():
S/registration/aligners/multi_aligner_impl.cpp:58-59,66,102; caller loop
S/registration/loop_detector/multi_loop_detector_brute_force_impl.cpp:64-133), so alignment k goes to rank
k mod G with NO collective on the data path.  The only exchange is ONE all-gather of the fixed-size result
records at the end so that every rank can apply the accept gates (:94-112).  Backend: torch.distributed
("nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""
import numpy as np

RECORD_FLOATS = 20  # X (12) + status, num_iterations, num_inliers, num_outliers, num_correspondences, chi_inliers, k, pad


def shard(K, world, rank):
    """indices of the alignments rank `rank` of `world` owns: k -> k mod world"""
    return list(range(rank, K, world))


def pack_record(k, result):
    r = np.zeros(RECORD_FLOATS, dtype=np.float64)
    X = np.asarray(result["moving_in_fixed"], dtype=np.float64).reshape(-1)
    r[:X.size] = X
    last = result["last"]
    r[12:19] = [result["status"], result["num_iterations"], last["num_inliers"], last["num_outliers"],
                last["num_correspondences"], last["chi_inliers"], k]
    return r


def unpack_record(r, tsize=12):
    shape = (3, 3) if tsize == 9 else (3, 4)
    return {"k": int(r[18]), "moving_in_fixed": r[:tsize].astype(np.float32).reshape(shape), "status": int(r[12]),
            "num_iterations": int(r[13]), "num_inliers": int(r[14]), "num_outliers": int(r[15]),
            "num_correspondences": int(r[16]), "chi_inliers": float(np.float32(r[17]))}


def all_gather_records(local_records, K, device=None):
    """local_records: list of pack_record() rows of this rank.  Returns the (K, RECORD_FLOATS) table in
    alignment order on every rank.  One collective, ~160 B per alignment (latency bound)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        table = np.zeros((K, RECORD_FLOATS))
        for r in local_records:
            table[int(r[18])] = r
        return table
    world = dist.get_world_size()
    per_rank = (K + world - 1) // world
    buf = torch.full((per_rank, RECORD_FLOATS), -1.0, dtype=torch.float64, device=device)
    for i, r in enumerate(local_records):
        buf[i] = torch.from_numpy(r).to(buf.device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    table = np.zeros((K, RECORD_FLOATS))
    for t in out:
        for row in t.cpu().numpy():
            if row[18] >= 0:
                table[int(row[18])] = row
    return table
