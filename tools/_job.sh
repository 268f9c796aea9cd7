#!/bin/bash
cd /root/repo
SRRG2_AMD_HOSTTIME=1 python tools/loop_compute.py 100000 8 2>&1 | tail -6
