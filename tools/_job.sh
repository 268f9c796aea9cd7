#!/bin/bash
# scratch job for gpurun
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/t34; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -k "projective or c3 or proj" 2>&1 | tail -4 > $O/pytest.txt
for m in 2 1 2 1; do
  echo "proj_ppt=$m" >> $O/ab.txt
  SRRG2_AMD_PROJ_PPT=$m python bench.py --workload c3 --no-cpu-baseline --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  c3', round(d['value']), d['ms_per_step'])" >> $O/ab.txt
done
