#!/bin/bash
cd /root/repo
for w in c2 c3 c4; do
timeout 300 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-all-cores --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'])"
done
SRRG2_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 50 --warmup 5 2>/dev/null | tail -1 | cut -c1-160
timeout 600 python -m pytest tests/test_multi_gpu_gloo.py -q 2>&1 | tail -1
