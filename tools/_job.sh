#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3z; mkdir -p $O; rm -f $O/ab2.txt
T="SRRG2_AMD_LDS_TILE=1 SRRG2_AMD_QUEUE_MIN=1000000000"
for w in "--workload c2" "--workload c2 --overlap 0.7" "--workload c2 --overlap 0.5" "--workload c2 --points 30000" "--workload c2 --points 200000" "--workload c2 --points 500000" "--workload c2 --points 200000 --overlap 0.6" "--workload c2 --points 1000000"; do
  bash tools/ab_env.sh $O/ab2.txt "$w" "-" "$T" "$T SRRG2_AMD_FAST_GATHER=1" "SRRG2_AMD_LDS_TILE=1"
done
cat $O/ab2.txt
