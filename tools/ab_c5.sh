#!/bin/bash
# A/B of environment settings on the C5 bench line.  usage: gpurun -- 'bash tools/ab_c5.sh <tag> "ENV=VAL ..." "ENV=VAL2" ...'
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); w=d.get('warm',{})
print('%-40s cold %.2f ms  kept %.2f  append %.2f  pcg %s (sum %d)' % ('$1',d['ms_per_step'],w.get('same_topology_ms',0),w.get('after_appending_one_variable_and_factor_ms',0),d['config']['pcg_iterations'],sum(d['config']['pcg_iterations'])))"; }
for rep in 1 2; do
for cfg in "$@"; do
  env $cfg python $R/bench.py --workload c5 --no-cpu-baseline 2>/dev/null | line "$cfg" | tee -a $O/ab.txt
done
done
