#!/usr/bin/env python
"""Aggregate a rocprofv3 PC-sampling CSV (host_trap / stochastic) by source line of a -gline-tables-only build:
usage: pc_summary.py <dir with *pc_sampling*.csv> <kernel substring> [top N]"""
import collections
import csv
import glob
import os
import sys

d, kern = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
files = [f for f in glob.glob(os.path.join(d, "**", "*.csv"), recursive=True) if "pc_sampling" in os.path.basename(f)]
print("files:", [(f, os.path.getsize(f)) for f in files])
for f in files:
    with open(f, newline="") as fh:
        rd = csv.reader(fh)
        header = next(rd)
        print("header:", header)
        rows = []
        for i, r in enumerate(rd):
            if i < 3:
                print("row:", r)
            rows.append(r)
    col = {h: i for i, h in enumerate(header)}
    ci = next((col[c] for c in col if c.lower() in ("instruction_comment", "instruction comment")), None)
    ii = next((col[c] for c in col if c.lower() == "instruction"), None)
    by_line, by_inst = collections.Counter(), collections.Counter()
    n = 0
    for r in rows:
        text = " ".join(r)
        if kern and kern not in text:
            # kernel name may not be in the row: keep everything if no row matches at all (handled below)
            pass
        n += 1
        if ci is not None:
            by_line[r[ci]] += 1
        if ii is not None:
            by_inst[r[ii].split()[0] if r[ii] else "?"] += 1
    print("samples:", n)
    print("== by source line")
    for k, v in by_line.most_common(top):
        print("%6d %5.1f%%  %s" % (v, 100.0 * v / max(n, 1), k))
    print("== by opcode")
    for k, v in by_inst.most_common(25):
        print("%6d %5.1f%%  %s" % (v, 100.0 * v / max(n, 1), k))
