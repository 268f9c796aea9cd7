#!/bin/bash
# scratch job for gpurun
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/t25; mkdir -p $O
run() { # label, env...
  echo "$1" >> $O/ab.txt
  shift
  for rep in 1 2; do
  env "$@" python bench.py --workload c4 --batch 256 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  c4_256', round(d['value']), d['ms_per_step'])" >> $O/ab.txt
  done
  env "$@" python bench.py --workload c4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  c4_32', round(d['value']), d['ms_per_step'])" >> $O/ab.txt
  env "$@" python bench.py --workload c2 --no-cpu-baseline --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  c2', round(d['value']), d['ms_per_step'])" >> $O/ab.txt
  env "$@" python tools/bench_tracker.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  tracker ms/frame', d['ms_per_frame_on_device'], 'compute', d['ms']['compute'])" >> $O/ab.txt
}
run "default (0.02)" X=1
for v in 0.005 0.01 0.04 0.08; do run "PAD_MIN=$v" SRRG2_AMD_LIB=$R/build_variants/lib_min$v.so; done
