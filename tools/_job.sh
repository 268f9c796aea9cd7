#!/bin/bash
cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 300 2>&1 | tail -3
python tools/loop_compute.py 100000 300
python tools/loop_compute.py 100000 300
python tools/loop_compute.py 10000 300
timeout 300 python bench.py --workload c4 --steps 100 --warmup 10 --no-cpu-all-cores --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4', d['value'])"
timeout 300 python bench.py --workload c3 --steps 200 --warmup 20 --no-cpu-all-cores --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', d['value'])"
export TMPDIR=/tmp
rm -rf /tmp/tr; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tr -o t -- python tools/loop_compute.py 100000 50 > /tmp/tr.log 2>&1
python tools/rocpd_summary.py /tmp/s.txt k=$(ls /tmp/tr/*.db | head -1); grep "k_icp_control" /tmp/s.txt | cut -c1-140
