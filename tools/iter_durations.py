#!/usr/bin/env python
"""per-dispatch durations [us] of the step kernels of ONE compute() from a rocprofv3 --kernel-trace rocpd db: the last
`n` dispatches (default 10 = one compute() of 10 iterations), and with a --pmc db the counters of those dispatches"""
import sqlite3
import sys

db, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10
cur = sqlite3.connect(db).cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
if "kernels" in tables:
    rows = list(cur.execute("select name, (end-start)/1000.0 from kernels where name like '%k_icp_step%' and name not like '%queue%' order by start"))
    print("durations_us", [round(r[1], 1) for r in rows[-n:]])
    print("kernels", sorted(set(r[0].split("(")[0].replace("void ", "") for r in rows[-n:])))
if "counters_collection" in tables:
    names = [r[0] for r in cur.execute("select distinct counter_name from counters_collection")]
    for c in names:
        rows = list(cur.execute("select value from counters_collection where counter_name = ? and kernel_name like '%k_icp_step%' and kernel_name not like '%queue%' order by dispatch_id", (c,)))
        print(c, [round(r[0]) for r in rows[-n:]])
