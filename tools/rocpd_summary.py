#!/usr/bin/env python
"""Summarise rocprofv3 rocpd (SQLite) outputs into a small text file for profiles/.

usage: rocpd_summary.py OUT.txt  label=path/to/results.db [label=...]
A db from `--kernel-trace --stats` yields the per-kernel duration table; a db from a `--pmc X`
pass yields per-kernel averages of every collected counter.
"""
import sqlite3
import sys


def main():
    out = open(sys.argv[1], "w")
    for arg in sys.argv[2:]:
        label, path = arg.split("=", 1)
        cur = sqlite3.connect(path).cursor()
        out.write("## %s (%s)\n" % (label, path))
        out.write("# kernel durations [us]: name | calls | total | average | percent\n")
        for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            out.write("%s | %d | %.3f | %.3f | %.2f\n" % (r[0], r[1], r[2], r[3], r[4]))
        # A pipelined batch runs a pass as several dispatches (one per part): the per-name average above then mixes part- and
        # whole-batch dispatches.  The step kernels again, by the number of alignments a dispatch covers (grid y).
        try:
            cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
            gy = next((c for c in ("grid_y", "grid_size_y") if c in cols), None)
            if gy:
                rows = list(cur.execute("select name, %s, count(*), avg(end - start) / 1000.0, min(end - start) / 1000.0, "
                                        "max(end - start) / 1000.0 from kernels where name like '%%k_icp_%%' group by name, %s "
                                        "order by name, %s" % (gy, gy, gy)))
                if len(set(r[0] for r in rows)) < len(rows):
                    out.write("# ICP kernels by alignments per dispatch: name | grid y | calls | average | min | max [us]\n")
                    for r in rows:
                        out.write("%s | %d | %d | %.3f | %.3f | %.3f\n" % (r[0].split("(")[0], r[1], r[2], r[3], r[4], r[5]))
        except sqlite3.Error:
            pass
        rows = list(cur.execute(
            "select kernel_name, counter_name, avg(value), min(value), max(value), count(*) from counters_collection "
            "group by kernel_name, counter_name order by kernel_name"))
        if rows:
            out.write("# counters per dispatch: kernel | counter | avg | min | max | dispatches\n")
            for r in rows:
                out.write("%s | %s | %.3f | %.3f | %.3f | %d\n" % r)
        out.write("\n")
    out.close()


if __name__ == "__main__":
    main()
