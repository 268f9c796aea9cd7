#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
O=$R/gpurun_out/r3t; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1
tail -2 $O/pytest_gpu.txt
cd /tmp
for w in c2 c3; do
  rm -rf /tmp/trk
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/trk -o t -- python $R/bench.py --workload $w --steps 30 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  python $R/tools/rocpd_summary.py /tmp/s.txt kernel_trace_stats=$(find /tmp/trk -name '*.db' | head -1) > /dev/null 2>&1
  echo "$w: $(grep '^k_icp_control' /tmp/s.txt | cut -d'|' -f1-4 | cut -c1-20,100-)"
done
cd $R
rm -f $O/ab.txt
for w in "--workload c2" "--workload c3" "--workload c4 --batch 32 --steps 10" "--workload c4 --batch 8 --steps 10"; do
  bash tools/ab_env.sh $O/ab.txt "$w" "-" "-"
done
cat $O/ab.txt
