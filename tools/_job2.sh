#!/bin/bash
cd /root/repo
python tools/loop_compute.py 100000 200
SRRG2_AMD_TUNE=33554432 python tools/loop_compute.py 100000 200
python tools/loop_compute.py 100000 200
SRRG2_AMD_TUNE=33554432 python tools/loop_compute.py 100000 200
