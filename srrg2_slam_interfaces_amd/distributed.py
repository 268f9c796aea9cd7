"""Multi-GPU sharding of independent alignments.

Loop-closure candidate alignments are mutually independent (the aligner state is fully reset per compute():
S/registration/aligners/multi_aligner_impl.cpp:58-59,66,102; caller loop
S/registration/loop_detector/multi_loop_detector_brute_force_impl.cpp:64-133), so alignment k goes to rank
k mod G with NO collective on the data path.  The only exchange is ONE collective over the fixed-size result
records at the end so that every rank can apply the accept gates (:94-112) and emit closures with their
information matrix.  Two equivalent forms (same table on every rank, bit for bit):
  * all-gather of the ranks' own rows;
  * all-reduce(sum) of a K-row table in which every rank fills its own rows and leaves the others zero -- the
    "all-reduce of the final Hessian" of BASELINE.json's north_star: the rows carry H (upper triangle) next to X and the
    statistics; the sum runs over the rows' int64 bit patterns (x + 0 = x for every bit pattern, -0.0 included), so it
    reproduces every row unchanged.
Backend: torch.distributed ("nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  The record layout is the
C ABI's (srrg2_multi_gpu_pack_record, include/srrg2_slam_amd.h); tests/test_multi_gpu_gloo.py checks both agree.
"""
import numpy as np

RECORD_FLOATS = 41  # X 12 | status, num_iterations, num_inliers, num_outliers, num_correspondences, chi_inliers, k, D | H upper 21
_K_SLOT = 18


def shard(K, world, rank):
    """indices of the alignments rank `rank` of `world` owns: k -> k mod world"""
    return list(range(rank, K, world))


def pack_record(k, result):
    r = np.zeros(RECORD_FLOATS, dtype=np.float64)
    X = np.asarray(result["moving_in_fixed"], dtype=np.float64).reshape(-1)
    r[:X.size] = X
    last = result["last"] or {"num_inliers": 0, "num_outliers": 0, "num_correspondences": 0, "chi_inliers": 0.0}
    D = 3 if X.size == 9 else 6
    r[12:20] = [result["status"], result["num_iterations"], last["num_inliers"], last["num_outliers"],
                result.get("num_correspondences", last["num_correspondences"]), last["chi_inliers"], k, D]
    H = result.get("information")
    if H is not None:
        H = np.asarray(H, dtype=np.float64).reshape(D, D)
        r[20:20 + D * (D + 1) // 2] = H[np.triu_indices(D)]
    return r


def unpack_record(r, tsize=12):
    shape = (3, 3) if tsize == 9 else (3, 4)
    D = 3 if tsize == 9 else 6
    H = np.zeros((D, D), np.float32)
    iu = np.triu_indices(D)
    H[iu] = r[20:20 + len(iu[0])].astype(np.float32)
    H = H + np.triu(H, 1).T
    return {"k": int(r[_K_SLOT]), "moving_in_fixed": r[:tsize].astype(np.float32).reshape(shape), "status": int(r[12]),
            "num_iterations": int(r[13]), "num_inliers": int(r[14]), "num_outliers": int(r[15]),
            "num_correspondences": int(r[16]), "chi_inliers": float(np.float32(r[17])), "information": H}


def _local_table(local_records, K):
    table = np.zeros((K, RECORD_FLOATS))
    for r in local_records:
        table[int(r[_K_SLOT])] = r
    return table


def all_reduce_records(local_records, K, device=None):
    """The (K, RECORD_FLOATS) table on every rank by ONE all-reduce(sum): every rank contributes its own rows, zeros
    elsewhere (K x 328 B: 84 kB at K = 256, latency bound on any link)."""
    import torch
    import torch.distributed as dist

    table = _local_table(local_records, K)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return table
    if device is None:  # (NCCL / RCCL reduces device tensors only)
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu"
    # The sum runs over the int64 BIT PATTERNS of the float64 rows: every row is non-zero on exactly one rank, so the
    # integer sum reproduces its bits exactly -- a float sum would turn -0.0 into +0.0 (-0.0 + 0.0 = +0.0) and could
    # differ from the all-gather form in the sign of a zero.
    t = torch.from_numpy(table.view(np.int64).copy()).to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().view(np.float64)


def all_gather_records(local_records, K, device=None):
    """local_records: list of pack_record() rows of this rank.  Returns the (K, RECORD_FLOATS) table in
    alignment order on every rank.  One collective, 328 B per alignment (latency bound)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return _local_table(local_records, K)
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
            if row[_K_SLOT] >= 0:
                table[int(row[_K_SLOT])] = row
    return table


def exchange_records(local_records, K, device=None, mode="all_reduce"):
    return (all_reduce_records if mode == "all_reduce" else all_gather_records)(local_records, K, device=device)


def point_shard_reducer(lib, group=None, through_host=None):
    """The ``reduce`` callable of MultiAligner.set_point_shard on torch.distributed: sums (int64) / maxima (uint32 bit
    patterns of non-negative floats, reduced as int32: same order) over the ranks of ``group``, in place on the
    aligner's device buffer.  With an RCCL (``nccl``) group the buffer is copied into a torch tensor on the aligner's
    stream, all-reduced under that stream (torch orders the collective after the copy and the copy back after the
    collective: no host wait), and copied back; with a host-side group (``gloo``: the CPU tests, two ranks on one GPU) it
    goes through pinned-size host arrays, synchronously.  ``lib`` = the loaded C library (``_capi.lib()``):
    srrg2_amd_memcpy does the raw copies."""
    import ctypes as C

    import torch
    import torch.distributed as dist

    backend = dist.get_backend(group)
    if through_host is None:
        through_host = backend != "nccl"
    lib.srrg2_amd_memcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]

    def reduce(op, ptr, count, stream):
        dtype, nbytes, red = (torch.int64, 8 * count, dist.ReduceOp.SUM) if op == 0 else (torch.int32, 4 * count, dist.ReduceOp.MAX)
        if through_host:
            if stream:
                torch.cuda.ExternalStream(stream).synchronize()
            host = torch.empty(count, dtype=dtype)
            if lib.srrg2_amd_memcpy(C.c_void_p(host.data_ptr()), C.c_void_p(ptr), nbytes, 0, None):
                raise RuntimeError("srrg2_amd_memcpy (device -> host) failed")
            dist.all_reduce(host, op=red, group=group)
            if lib.srrg2_amd_memcpy(C.c_void_p(ptr), C.c_void_p(host.data_ptr()), nbytes, 1, None):
                raise RuntimeError("srrg2_amd_memcpy (host -> device) failed")
            return
        ext = torch.cuda.ExternalStream(stream)
        with torch.cuda.stream(ext):
            buf = torch.empty(count, dtype=dtype, device="cuda")
            if lib.srrg2_amd_memcpy(C.c_void_p(buf.data_ptr()), C.c_void_p(ptr), nbytes, 2, C.c_void_p(stream)):
                raise RuntimeError("srrg2_amd_memcpy (device -> device) failed")
            dist.all_reduce(buf, op=red, group=group)  # (ordered on `ext`: c10d waits for and signals the current stream)
            if lib.srrg2_amd_memcpy(C.c_void_p(ptr), C.c_void_p(buf.data_ptr()), nbytes, 2, C.c_void_p(stream)):
                raise RuntimeError("srrg2_amd_memcpy (device -> device) failed")
            buf.record_stream(ext)

    return reduce
