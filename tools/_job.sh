#!/bin/bash
cd /root/repo
for n in 10000 30000 100000; do
python tools/loop_compute.py $n 200
SRRG2_AMD_W1_MAX=1000000 python tools/loop_compute.py $n 200
done
SRRG2_AMD_W1_MAX=1000000 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 300 2>&1 | tail -1
