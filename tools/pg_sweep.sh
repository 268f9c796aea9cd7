#!/bin/bash
# knob sweep of the C5 solve (structure kept: the third of three solves of tools/bench_posegraph.py)
#   usage: gpurun -- 'bash tools/pg_sweep.sh <tag> "ENV=VAL ENV2=VAL" "ENV=VAL2" ...'
TAG=${1:-sweep}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
for cfg in "$@"; do
  env $cfg python $R/tools/bench_posegraph.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-60s %.1f ms  %d its %s chi %.4f' % ('$cfg', d['gpu_solve_s']*1e3, sum(d['gpu_pcg_its']), d['gpu_pcg_its'], d['gpu_chi'][-1]))" | tee -a $O/sweep.txt
done
