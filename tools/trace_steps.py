#!/usr/bin/env python
"""print per-dispatch durations [us] of k_icp_step / k_icp_control from a rocprofv3 rocpd db"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
for pat in ("%k_icp_step%", "%k_icp_control%"):
    rows = [r[0] / 1000 for r in cur.execute("select (end-start) from kernels where name like ? order by start", (pat,))]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    print(pat, len(rows), "avg %.1f" % (sum(rows) / max(len(rows), 1)), [round(x, 1) for x in rows[-n:]])
