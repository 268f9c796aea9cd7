#!/usr/bin/env python
"""per-(kernel, grid size) durations of the pose-graph kernels from a rocprofv3 --kernel-trace rocpd db (the grid size tells
the level and phase of a k_mg_op launch apart)"""
import collections
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
q = "select name, %s, (end-start)/1000.0 from kernels" % (gx or "0")
agg = collections.defaultdict(list)
for name, g, d in cur.execute(q):
    short = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    agg[(short.split("(")[0], g)].append(d)
tot = sum(sum(v) for v in agg.values())
print("columns:", cols)
for (name, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-28s grid %8s  calls %6d  total %9.1f us (%4.1f%%)  avg %7.2f  min %7.2f" % (name, g, len(v), sum(v), 100.0 * sum(v) / tot, sum(v) / len(v), min(v)))
