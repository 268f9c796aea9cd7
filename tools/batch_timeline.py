#!/usr/bin/env python
"""Kernel timeline of one pipelined compute_batch() in a rocprofv3 --kernel-trace rocpd db (two streams): every kernel with
its stream, start offset and duration; the union of the busy intervals against the span.
The batch shown: the last one whose kernels run on two streams (bench.py's final batches run with the profiling events on and
are not pipelined).   usage: batch_timeline.py <db>"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
sid = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
rows = [r for r in cur.execute("select name, start, end, %s from kernels order by start" % (sid or "0"))]
finals = [i for i, r in enumerate(rows) if "k_icp_control_final" in r[0] or "k_icp_final_wave" in r[0]]  # (the last step of a part)
# a pipelined batch ends with TWO final control steps on different streams; find the last such pair
pair = None
for a, b in zip(finals, finals[1:]):
    if rows[a][3] != rows[b][3] and rows[b][1] - rows[a][2] < 200000:  # (ns: the two halves end close together)
        pair = (a, b)
if pair is None:
    print("no pipelined batch in this trace")
    sys.exit(0)
end_i = pair[1]
prev_finals = [i for i in finals if i < pair[0]]
begin_i = (prev_finals[-1] + 1) if prev_finals else 0
seq = [r for r in rows[begin_i:end_i + 1] if not r[0].startswith("__amd")]
t0 = seq[0][1]
ivals = []
for n, s, e, st in seq:
    short = n.replace("void ", "").replace("(anonymous namespace)::", "").replace("srrg2amd::", "").split("(")[0]
    print("%9.2f us  %8.2f us  stream %s  %s" % ((s - t0) / 1000.0, (e - s) / 1000.0, st, short))
    ivals.append((s, e))
ivals.sort()
busy, cs, ce = 0, ivals[0][0], ivals[0][1]
for s, e in ivals[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
span = max(e for _, e in ivals) - t0
print("span %.2f us, some kernel running %.2f us (%.1f %%), sum of kernel durations %.2f us" %
      (span / 1000.0, busy / 1000.0, 100.0 * busy / span, sum(e - s for s, e in ivals) / 1000.0))
