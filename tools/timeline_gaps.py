#!/usr/bin/env python
"""Kernel timeline of the LAST compute() in a rocprofv3 --kernel-trace rocpd db: start offset, duration and the idle gap in
front of every kernel between k_icp_init and the final control step.   usage: timeline_gaps.py <db> [kernels per compute]"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = [(s, e, n) for n, s, e in cur.execute("select name, start, end from kernels order by start")]
# the last compute(): from the last k_icp_init on
# (round 6, late: from a handle's second compute() on the prologue rides in the first pass -- k_icp_step_cnl_init / k_icp_step_fused_init)
last = max(i for i, r in enumerate(rows) if r[2].startswith("k_icp_init") or "_init<" in r[2])
seq = rows[last:]
t0 = seq[0][0]
prev_end = None
tot_gap = 0.0
for s, e, n in seq:
    gap = 0.0 if prev_end is None else (s - prev_end) / 1000.0
    tot_gap += gap
    short = n.replace("void ", "").split("(")[0]
    print("%8.2f us  +%6.2f gap  %7.2f us  %s" % ((s - t0) / 1000.0, gap, (e - s) / 1000.0, short))
    prev_end = e
    if "control_final" in n or "final_wave" in n or "k_icp_finalize" in n or "k_icp_small" in n:
        break
print("span %.2f us, kernels %.2f us, gaps %.2f us" % ((prev_end - t0) / 1000.0, (prev_end - t0) / 1000.0 - tot_gap, tot_gap))
