#!/bin/bash
# A/B of two builds of the library on the C4 / C2 bench lines.  usage: gpurun -- 'bash tools/ab_lib.sh <tag> <alt .so> [reps]'
TAG=$1; ALT=$2; REPS=${3:-2}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-10s %-6s value %.0f  ms_per_step %.4f  frac %.4f' % ('$1','$2',d['value'],d['ms_per_step'],d['roofline']['frac']))"; }
for r in $(seq $REPS); do
  for which in base alt; do
    if [ $which = alt ]; then export SRRG2_AMD_LIB=$R/$ALT; else unset SRRG2_AMD_LIB; fi
    python $R/bench.py --workload c4 --batch 256 --no-cpu-baseline 2>/dev/null | line c4_256 $which | tee -a $O/ab.txt
    python $R/bench.py --workload c4 --no-cpu-baseline 2>/dev/null | line c4_32 $which | tee -a $O/ab.txt
    python $R/bench.py --workload c2 --no-cpu-baseline 2>/dev/null | line c2 $which | tee -a $O/ab.txt
  done
done
