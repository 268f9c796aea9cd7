#!/bin/bash
# scratch job for gpurun
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/t33; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_tuning.py tests/test_gpu_full_size.py -x -q 2>&1 | tail -4 > $O/pytest.txt
run() { # label, env...
  echo "$1" >> $O/ab.txt
  shift
  for rep in 1 2; do
  env "$@" python bench.py --workload c4 --batch 256 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  c4_256', round(d['value']), d['ms_per_step'])" >> $O/ab.txt
  done
  env "$@" python bench.py --workload c4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  c4_32', round(d['value']), d['ms_per_step'])" >> $O/ab.txt
  env "$@" python bench.py --workload c2 --no-cpu-baseline --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  c2', round(d['value']), d['ms_per_step'])" >> $O/ab.txt
  env "$@" python bench.py --workload c2 --overlap 0.6 --no-cpu-baseline --steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  c2 overlap 0.6', round(d['value']), d['ms_per_step'])" >> $O/ab.txt
}
run "re-test on" X=1
run "re-test off" SRRG2_AMD_LIB=$R/build_variants/lib_noretest.so
