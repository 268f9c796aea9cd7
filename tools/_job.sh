#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 300 > gpurun_out/t_gpu.log 2>&1
tail -3 gpurun_out/t_gpu.log
python tools/loop_compute.py 100000 200
SRRG2_AMD_TUNE=8388608 python tools/loop_compute.py 100000 200
python tools/loop_compute.py 100000 200
timeout 300 python bench.py --workload c4 --steps 100 --warmup 10 --no-cpu-all-cores --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4', d['value'])"
