#!/bin/bash
cd /root/repo
for i in 1 2; do
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-all-cores --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench c2', d['value'], d['ms_per_step'])"
done
timeout 300 python bench.py --workload c3 --steps 200 --warmup 20 --no-cpu-all-cores --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench c3', d['value'], d['ms_per_step'])"
