#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$R/gpurun_out/r4j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tuning.py -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
bash tools/ab_env.sh $O/ab_c2.txt "--workload c2 --steps 200 --warmup 20" "-" "SRRG2_AMD_SEARCH_TEAM=4" "-" "SRRG2_AMD_SEARCH_TEAM=4"
bash tools/ab_env.sh $O/ab_c2.txt "--workload c2 --overlap 0.6 --steps 100 --warmup 20" "SRRG2_AMD_SEARCH_LISTS=0" "-"
bash tools/ab_env.sh $O/ab_c2.txt "--workload c4 --batch 256 --steps 10 --warmup 2" "-" "-"
bash tools/ab_env.sh $O/ab_c2.txt "--workload c4 --batch 32 --steps 30 --warmup 3" "-" "-"
cat $O/ab_c2.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_c2 -o t -- python $R/bench.py --workload c2 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/trace_steps.py $(find /tmp/tr_c2 -name '*.db' | head -1) > $O/trace_c2_steps.txt 2>/dev/null
head -40 $O/trace_c2_steps.txt
