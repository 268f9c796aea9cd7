#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
rm -rf /tmp/tr; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tr -o t -- python bench.py --points 1000000 --steps 10 --warmup 2 --no-cpu-all-cores --no-cpu-baseline > /tmp/tr.log 2>&1
python tools/rocpd_summary.py /tmp/s.txt k=$(ls /tmp/tr/*.db | head -1); cut -c1-120 /tmp/s.txt | head -12
python tools/trace_steps.py $(ls /tmp/tr/*.db | head -1) | cut -c1-330
