#!/bin/bash
cd /root/repo
for i in 1 2 3; do
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-all-cores --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', d['value'], d['roofline']['avg_launch_ms'])"
done
git stash -q 2>/dev/null
