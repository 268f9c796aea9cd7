#!/bin/bash
cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 300 2>&1 | tail -2
python tools/loop_compute.py 100000 300
python tools/loop_compute.py 100000 300
python tools/loop_compute.py 10000 300
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-all-cores --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench c2', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
timeout 300 python bench.py --workload c4 --steps 100 --warmup 10 --no-cpu-all-cores --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4', d['value'])"
