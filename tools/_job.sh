#!/bin/bash
cd /root/repo
for n in 300000 1000000; do
for c in 0 3 4 5 6 8; do
echo -n "RMAX_CAP=$c  "; SRRG2_AMD_RMAX_CAP=$c python tools/loop_compute.py $n 15
done
for t in 4 16 32; do
echo -n "CELL_TARGET=$t  "; SRRG2_AMD_CELL_TARGET=$t python tools/loop_compute.py $n 15
done
done
