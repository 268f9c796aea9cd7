#!/bin/bash
cd /root/repo
SRRG2_AMD_TUNE=16777216 timeout 300 python tools/loop_compute.py 100000 0 2>&1 | grep "^it " | sort -k4n -k2n | uniq | head -60
