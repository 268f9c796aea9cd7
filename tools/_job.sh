#!/bin/bash
# scratch job for gpurun
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/t17; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_tuning.py tests/test_gpu_full_size.py -x -q 2>&1 | tail -5 > $O/pytest.txt
python bench.py --workload c4 --batch 256 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4_256', d['value'], d['ms_per_step'])" >> $O/ab.txt
