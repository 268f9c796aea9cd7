#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
rm -rf /tmp/tr; timeout 300 rocprofv3 --kernel-trace -d /tmp/tr -o t -- python bench.py --workload c4 --steps 5 --warmup 1 --no-cpu-all-cores --no-cpu-baseline > /tmp/tr.log 2>&1
python tools/trace_steps.py $(ls /tmp/tr/*.db | head -1) | cut -c1-400
