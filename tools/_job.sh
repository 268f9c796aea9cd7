#!/bin/bash
# scratch job for gpurun
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/t19; mkdir -p $O
for m in 0 1; do
  echo "stagger=$m" >> $O/ab.txt
  for rep in 1 2; do
  SRRG2_AMD_STAGGER=$m python bench.py --workload c4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4_32', d['value'], d['ms_per_step'])" >> $O/ab.txt
  done
  SRRG2_AMD_STAGGER=$m python bench.py --workload c4 --batch 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4_8', d['value'], d['ms_per_step'])" >> $O/ab.txt
  SRRG2_AMD_STAGGER=$m python bench.py --workload c4 --batch 256 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4_256', d['value'], d['ms_per_step'])" >> $O/ab.txt
done
python -m pytest tests/test_gpu_tuning.py tests/test_gpu_full_size.py -x -q 2>&1 | tail -3 > $O/pytest.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d /tmp/tr32 -o t -- python $R/bench.py --workload c4 --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/batch_timeline.py $(find /tmp/tr32 -name '*.db' | head -1) > $O/timeline_c4_32.txt 2>&1
