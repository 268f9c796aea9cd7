#!/usr/bin/env python
"""Registers / LDS / scratch / occupancy per kernel, from `hipcc -Rpass-analysis=kernel-resource-usage` remarks.
  usage: hipcc <flags of csrc/Makefile> -Rpass-analysis=kernel-resource-usage -c kernels.hip -o /tmp/k.o 2> res.txt
         python tools/kernel_resources.py res.txt [substring ...]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2:]
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
names = [b.split()[0] for b in blocks]
try:
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
except Exception:
    dem = names
for b, d in zip(blocks, dem):
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    d = re.sub(r"^void ", "", d)
    d = re.sub(r"\(.*", "", d)[:78]
    if want and not any(w in d for w in want):
        continue
    print("%-80s vgpr %3d agpr %3d sgpr %3d scratch %4d occ %d lds %6d" % (
        d, g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
        g(r"LDS Size \[bytes/block\]")))
