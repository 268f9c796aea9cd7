#!/usr/bin/env python
"""GPU time of a compute(): first kernel start to last kernel end, from a rocprofv3 --kernel-trace rocpd db, over the compute() calls
that consist of k_icp_* kernels only (steady state: no set_fixed / list build in between).  usage: compute_span.py <db>"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = [(s, e, n.replace("void ", "").replace("srrg2amd::", "").replace("(anonymous namespace)::", "").split("(")[0])
        for n, s, e in cur.execute("select name, start, end from kernels order by start")]
ends = [i for i, r in enumerate(rows) if "final" in r[2]]
spans, seqs = [], []
for a, b in zip(ends[:-1], ends[1:]):
    sq = rows[a + 1:b + 1]
    if all(r[2].startswith("k_icp_") for r in sq):
        spans.append((sq[-1][1] - sq[0][0]) / 1000.0)
        seqs.append(sq)
if not spans:
    sys.exit("no steady-state compute() found")
sq = seqs[-1]
for s, e, n in sq:
    print("%8.2f us %7.2f us  %s" % ((s - sq[0][0]) / 1000.0, (e - s) / 1000.0, n[:70]))
spans.sort()
print("%d steady-state compute() calls: span median %.2f us, min %.2f, mean %.2f" % (len(spans), spans[len(spans) // 2], spans[0], sum(spans) / len(spans)))
