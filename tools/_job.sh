#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r3o
O=gpurun_out/r3o
SRRG2_AMD_ASYNC_CONTROL=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_given_correspondences.py tests/test_sensor_in_robot.py tests/test_reference_scenarios.py -m gpu -x -q > $O/pytest_async2.log 2>&1
tail -3 $O/pytest_async2.log
grep -c "timed out" $O/pytest_async2.log
rm -f $O/ab2.txt
for w in "--workload c2" "--workload c3" "--workload c4 --batch 8 --steps 10" "--workload c4 --batch 32 --steps 10"; do
  timeout 600 bash tools/ab_env.sh $O/ab2.txt "$w" "SRRG2_AMD_ASYNC_CONTROL=0" "SRRG2_AMD_ASYNC_CONTROL=1" "SRRG2_AMD_ASYNC_CONTROL=0" "SRRG2_AMD_ASYNC_CONTROL=1"
done
cat $O/ab2.txt
SRRG2_AMD_ASYNC_CONTROL=1 python bench.py --workload c2 --no-cpu-baseline 2>&1 | grep -c "timed out"
