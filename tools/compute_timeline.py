#!/usr/bin/env python
"""The kernels (and fills / copies) of the LAST compute() in a rocprofv3 --kernel-trace rocpd db -- from the kernel behind the previous
final step to this one's -- with their gaps.   usage: compute_timeline.py <db>"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = [(s, e, n.replace("void ", "").replace("srrg2amd::", "").replace("(anonymous namespace)::", "").split("(")[0])
        for n, s, e in cur.execute("select name, start, end from kernels order by start")]
ends = [i for i, r in enumerate(rows) if "final" in r[2] or "k_icp_finalize" in r[2]]
sq = rows[ends[-2] + 1:ends[-1] + 1]
t0, prev = sq[0][0], sq[0][0]
for s, e, n in sq:
    print("%8.1f +%5.1f %7.1f us  %s" % ((s - t0) / 1000, (s - prev) / 1000, (e - s) / 1000, n[:64]))
    prev = e
print("span %.1f us, kernels %.1f us" % ((sq[-1][1] - t0) / 1000, sum(e - s for s, e, _ in sq) / 1000))
