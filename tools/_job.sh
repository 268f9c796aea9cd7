#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
O=$R/gpurun_out/r3u; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1
tail -2 $O/pytest_gpu.txt
rm -f $O/ab.txt
for w in "--workload c2" "--workload c3" "--workload c4 --batch 32 --steps 10" "--workload c4 --batch 8 --steps 10"; do
  bash tools/ab_env.sh $O/ab.txt "$w" "-" "-" "-"
done
cat $O/ab.txt
