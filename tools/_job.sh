#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$R/gpurun_out/r4n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tuning.py tests/test_reference_scenarios.py tests/test_sensor_in_robot.py -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
bash tools/ab_env.sh $O/ab.txt "--workload c2 --steps 200 --warmup 20" "-" "-"
bash tools/ab_env.sh $O/ab.txt "--workload c3 --steps 100 --warmup 10" "-"
bash tools/ab_env.sh $O/ab.txt "--workload c4 --batch 32 --steps 30 --warmup 3" "-"
bash tools/ab_env.sh $O/ab.txt "--workload c4 --batch 8 --steps 30 --warmup 3" "-"
bash tools/ab_env.sh $O/ab.txt "--workload c4 --batch 256 --steps 10 --warmup 2" "-"
cat $O/ab.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_c2 -o t -- python $R/bench.py --workload c2 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/trace_steps.py $(find /tmp/tr_c2 -name '*.db' | head -1) > $O/trace_c2_steps.txt 2>/dev/null
head -40 $O/trace_c2_steps.txt
python $R/tools/rocpd_summary.py $O/rocprofv3_c2_summary.txt kernel_trace_stats=$(find /tmp/tr_c2 -name '*.db' | head -1); head -9 $O/rocprofv3_c2_summary.txt | cut -c1-150
