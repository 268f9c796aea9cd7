#!/bin/bash
# A/B of environment settings on the C2 / C4 bench lines.  usage: gpurun -- 'bash tools/ab_env.sh <tag> "ENV=VAL ..." "ENV=VAL2" ...'
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-44s %-7s value %8.0f  ms_per_step %.4f' % ('$1','$2',d['value'],d['ms_per_step']))"; }
for cfg in "$@"; do
  env $cfg python $R/bench.py --workload c2 --no-cpu-baseline 2>/dev/null | line "$cfg" c2 | tee -a $O/ab.txt
  env $cfg python $R/bench.py --workload c4 --no-cpu-baseline 2>/dev/null | line "$cfg" c4_32 | tee -a $O/ab.txt
  env $cfg python $R/bench.py --workload c4 --batch 256 --no-cpu-baseline 2>/dev/null | line "$cfg" c4_256 | tee -a $O/ab.txt
done
