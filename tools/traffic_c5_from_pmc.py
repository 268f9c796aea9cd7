#!/usr/bin/env python
"""HBM-side traffic of the pose-graph solve (C5) from three rocprofv3 --pmc passes of tools/bench_posegraph.py
(FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU + SQ_WAVES; separate passes, SRRG2_AMD_PG_GRAPH=0 so that every kernel is a
dispatch of its own), summarised by tools/rocpd_summary.py.  Corrections of MI355X_MICROARCH.md: counters in KiB;
on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes -> doubled; WRITE_SIZE taken as is.

usage: traffic_c5_from_pmc.py OUT.json fetch_summary.txt write_summary.txt valu_summary.txt trace_summary.txt SOLVES GN PCG"""
import json
import sys


def counters(path, name):
    out = {}
    for line in open(path):
        f = [x.strip() for x in line.split("|")]
        if len(f) == 6 and f[1] == name:
            k = f[0].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            out[k] = (float(f[2]), int(f[5]))
    return out


def durations(path):
    out = {}
    for line in open(path):
        f = [x.strip() for x in line.split("|")]
        if len(f) == 5 and f[1].isdigit():
            k = f[0].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            out[k] = (float(f[3]), int(f[1]))
    return out


def main():
    out, fp, wp, vp, tp = sys.argv[1:6]
    solves, gn, pcg = int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])
    f, w = counters(fp, "FETCH_SIZE"), counters(wp, "WRITE_SIZE")
    v, nw = counters(vp, "SQ_INSTS_VALU"), counters(vp, "SQ_WAVES")
    d = durations(tp)
    kernels, total = {}, 0.0
    for k in sorted(set(f) | set(w)):
        if k.startswith("__amd"):
            continue
        fb = 2.0 * 1024.0 * f.get(k, (0.0, 0))[0]
        wb = 1024.0 * w.get(k, (0.0, 0))[0]
        n = f.get(k, w.get(k))[1]
        us = d.get(k, (0.0, 0))[0]
        kernels[k] = {"dispatches_per_solve": n / solves, "fetch_bytes_corrected": fb, "write_bytes": wb, "avg_us": us,
                      "bytes_per_s": (fb + wb) / (us * 1e-6) if us > 0 else None,
                      "frac_of_8TBs": (fb + wb) / (us * 1e-6) / 8e12 if us > 0 else None,
                      "valu_per_wave": v[k][0] / nw[k][0] if k in v and k in nw and nw[k][0] > 0 else None,
                      "us_per_solve": us * n / solves}
        total += (fb + wb) * n / solves
    dom = max(kernels, key=lambda k: kernels[k]["us_per_solve"])
    json.dump({"workload": "c5", "bytes_per_solve": total, "bytes_per_gn_iteration": total / gn,
               "gn_iterations": gn, "pcg_iterations": pcg, "dominant_kernel": dom,
               "dominant_kernel_frac_of_8TBs": kernels[dom]["frac_of_8TBs"], "kernels": kernels,
               "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU SQ_WAVES in separate passes of "
                         "tools/bench_posegraph.py (3 solves, HIP graph off), averaged per dispatch; KiB -> bytes; FETCH_SIZE "
                         "doubled (gfx950 tallies 128-byte requests at 64 bytes); includes Infinity-Cache hits; durations from "
                         "a --kernel-trace pass of the same command"},
              open(out, "w"), indent=1)
    print(json.dumps({k: (round(x["us_per_solve"]), round(x["frac_of_8TBs"] or 0, 3)) for k, x in kernels.items()}))
    print("total GB per solve", total / 1e9, "dominant", dom, kernels[dom]["frac_of_8TBs"])


if __name__ == "__main__":
    main()
