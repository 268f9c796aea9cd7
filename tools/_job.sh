#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t29
for k in 1 2 3 4; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['step_ms_min_median_max'], d.get('value_long'))" >> gpurun_out/t29/driver_like.txt; done
