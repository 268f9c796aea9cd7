#!/bin/bash
# knob sweep on a larger graph.  usage: gpurun -- 'bash tools/pg_sweep_big.sh <tag> V E "ENV=VAL ..." ...'
TAG=$1; V=$2; E=$3; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
for cfg in "$@"; do
  env $cfg timeout 300 python $R/tools/bench_posegraph.py $V $E 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-70s %.1f ms  %d its %s chi %.4g -> %.4g' % ('$cfg', d['gpu_solve_s']*1e3, sum(d['gpu_pcg_its']), d['gpu_pcg_its'], d['gpu_chi'][0], d['gpu_chi'][-1]))" | tee -a $O/sweep_big.txt
done
