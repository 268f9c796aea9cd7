#!/usr/bin/env python
"""HBM-side traffic per launch of the step kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate
passes: they do not fit one, MI355X_MICROARCH.md "HBM" and counter table).  Corrections of that guide: the counters are
in KiB; on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes -> doubled; WRITE_SIZE is uncalibrated for
scattered stores and taken as is; Infinity-Cache hits are counted (memory-side of L2, not DRAM).

usage: traffic_from_pmc.py OUT.json WORKLOAD fetch.db write.db"""
import json
import sqlite3
import sys

KERNELS = ("k_icp_step<", "k_icp_step_tile<", "k_icp_step_cnl<", "k_icp_step_fast<", "k_icp_step_queue<", "k_icp_step_proj", "k_proj_zbuf")  # (k_icp_step_proj_fused matches k_icp_step_proj)


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for name, val, n in cur.execute("select kernel_name, avg(value), count(*) from counters_collection where "
                                    "counter_name = ? group by kernel_name", (counter,)):
        if any(k in name for k in KERNELS):
            out[name.split("(")[0].replace("void ", "")] = (val, n)
    return out


def main():
    out, workload, fdb, wdb = sys.argv[1:5]
    per_launch = int(sys.argv[5]) if len(sys.argv) > 5 else None  # C4: alignments per launch of the profiled command
    f, w = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    kernels = {}
    total = 0.0
    passes = 0
    for k in sorted(set(f) | set(w)):
        fb = 2.0 * 1024.0 * f.get(k, (0.0, 0))[0]
        wb = 1024.0 * w.get(k, (0.0, 0))[0]
        n = f.get(k, (0, 0))[1]
        kernels[k] = {"fetch_bytes_corrected": fb, "write_bytes": wb, "dispatches": n}
        total += (fb + wb) * n
        if "queue" not in k and "zbuf" not in k:
            passes += n  # one step-kernel dispatch (k_icp_step, k_icp_step_tile, k_icp_step_cnl or k_icp_step_fast) per slice pass; the deferred-search kernel runs in some
    total = total / max(passes, 1)
    json.dump({"workload": workload, "bytes_per_slice_pass": total, "slice_passes": passes, "kernels": kernels,
               **({"alignments_per_launch": per_launch} if per_launch else {}),
               "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, averaged per dispatch; KiB -> bytes; "
                         "FETCH_SIZE doubled (gfx950 tallies 128-byte requests at 64 bytes); includes Infinity-Cache hits"},
              open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
