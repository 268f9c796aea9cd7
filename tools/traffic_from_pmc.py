#!/usr/bin/env python
"""HBM-side traffic per launch of the step kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate
passes: they do not fit one, MI355X_MICROARCH.md "HBM" and counter table).  Corrections of that guide: the counters are
in KiB; on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes -> doubled; WRITE_SIZE is uncalibrated for
scattered stores and taken as is; Infinity-Cache hits are counted (memory-side of L2, not DRAM).

A PASS is one finder + factor pass over ALL alignments of a compute_batch.  A pipelined batch runs a pass as several
dispatches (one per part, on its own stream: aligner_host.hip), so dispatches are weighted by the share of the batch they
cover (its problem dimension of the grid / alignments per launch) -- round 4 averaged per DISPATCH and mixed half- with whole-batch dispatches
(VERDICT r4 "weak" #2: 0.91x algorithmic reported where the per-pass file said 1.37x).  With the alignments per launch given
(C4) the output also lists the passes of the LAST compute() one by one when every dispatch of it covers the whole batch
(profile with SRRG2_AMD_BATCH_PIPELINE=0).

usage: traffic_from_pmc.py OUT.json WORKLOAD fetch.db write.db [alignments_per_launch [algorithmic_bytes_per_pass]]"""
import json
import sqlite3
import sys

KERNELS = ("k_icp_step<", "k_icp_step_tile<", "k_icp_step_cnl<", "k_icp_step_cnl_init<", "k_icp_step_fused", "k_icp_step_fast", "k_icp_step_queue<",
           "k_icp_step_proj", "k_proj_zbuf")  # (k_icp_step_proj_fused matches k_icp_step_proj; k_icp_step_fused_init matches k_icp_step_fused)


def dispatches(db, counter):
    """[(short kernel name, value, grid_size_y)] of the step kernels, in dispatch order"""
    cur = sqlite3.connect(db).cursor()
    out = []
    for name, val, gx, gy, wx in cur.execute("select kernel_name, value, grid_size_x, grid_size_y, workgroup_size_x from "
                                             "counters_collection where counter_name = ? order by dispatch_id", (counter,)):
        if any(k in name for k in KERNELS):
            short = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("srrg2amd::", "").split("(")[0]
            # alignments the dispatch covers: grid y -- or grid x (in workgroups) for the launches with fused control steps, which
            # walk the problems first (k_icp_step_cnl<..., true> / k_icp_step_fast<..., true>: x = problem, y = tile)
            # (round 6, late: the last template argument is 0 / 1 / 2 -- 1, 2 = fused --; the first pass of a single alignment with
            # compute()'s prologue inside, k_icp_step_cnl_init / k_icp_step_fused_init, and the fused grid kernel always walk x)
            s_ = short.rstrip()
            fused = (s_.startswith(("k_icp_step_cnl<", "k_icp_step_fast")) and s_.endswith((", true>", ", 1>", ", 2>"))) or \
                    s_.startswith(("k_icp_step_cnl_init<", "k_icp_step_fused"))
            out.append((short, float(val), int(gx) // max(int(wx), 1) if fused else int(gy)))
    return out


def per_kernel(rows):
    out = {}
    for name, val, gy in rows:
        t = out.setdefault(name, [0.0, 0, 0])
        t[0] += val
        t[1] += 1
        t[2] += gy
    return out


def main():
    out, workload, fdb, wdb = sys.argv[1:5]
    per_launch = int(sys.argv[5]) if len(sys.argv) > 5 else None  # C4: alignments per launch of the profiled command
    alg = float(sys.argv[6]) if len(sys.argv) > 6 else None
    fr, wr = dispatches(fdb, "FETCH_SIZE"), dispatches(wdb, "WRITE_SIZE")
    f, w = per_kernel(fr), per_kernel(wr)
    kernels = {}
    total = 0.0
    passes = 0.0
    for k in sorted(set(f) | set(w)):
        fsum, n, gy = f.get(k, (0.0, 0, 0))
        wsum = w.get(k, (0.0, 0, 0))[0]
        fb, wb = 2.0 * 1024.0 * fsum, 1024.0 * wsum  # (sums over all dispatches of the kernel)
        # one step-kernel dispatch per slice pass (k_icp_step, _tile, _cnl or _fast) -- or one per PART of a pipelined batch:
        # a dispatch counts for the share of the batch it covers; the deferred-search / z-buffer kernels add bytes, no passes
        share = (gy / float(per_launch)) if per_launch else float(n)
        kernels[k] = {"fetch_bytes_corrected_per_dispatch": fb / max(n, 1), "write_bytes_per_dispatch": wb / max(n, 1),
                      "dispatches": n, "passes": share}
        total += fb + wb
        if "queue" not in k and "zbuf" not in k:
            passes += share
    total = total / max(passes, 1e-9)
    doc = {"workload": workload, "bytes_per_slice_pass": total, "slice_passes": passes, "kernels": kernels,
           **({"alignments_per_launch": per_launch} if per_launch else {}),
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; bytes summed over all step-kernel dispatches "
                     "/ passes, a dispatch counting for the share of the batch it covers (grid_size_y / alignments per "
                     "launch); KiB -> bytes; FETCH_SIZE doubled (gfx950 tallies 128-byte requests at 64 bytes); includes "
                     "Infinity-Cache hits"}
    if alg:
        doc["algorithmic_bytes_per_pass"] = alg
        doc["ratio_to_algorithmic"] = total / alg
    if per_launch:  # the passes of the last compute(), one by one (whole-batch dispatches only)
        fl = [r for r in fr if "queue" not in r[0] and "zbuf" not in r[0]][-10:]
        wl = [r for r in wr if "queue" not in r[0] and "zbuf" not in r[0]][-10:]
        if len(fl) == len(wl) == 10 and all(r[2] == per_launch for r in fl + wl):
            doc["last_compute_passes"] = [
                {"pass": i, "kernel": a[0], "fetch_MB": 2.0 * 1024.0 * a[1] / 1e6, "write_MB": 1024.0 * b[1] / 1e6,
                 "total_MB": (2.0 * 1024.0 * a[1] + 1024.0 * b[1]) / 1e6,
                 **({"ratio_to_algorithmic": (2.0 * 1024.0 * a[1] + 1024.0 * b[1]) / alg} if alg else {})}
                for i, (a, b) in enumerate(zip(fl, wl))]
    json.dump(doc, open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
