#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$R/gpurun_out/r4e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_tuning.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_parity.txt 2>&1
tail -3 $O/pytest_parity.txt
for w in "--workload c4 --batch 256 --steps 10 --warmup 2" "--workload c4 --batch 32 --steps 30 --warmup 3" "--workload c2 --steps 200 --warmup 20"; do
  bash tools/ab_env.sh $O/ab_lists.txt "$w" "SRRG2_AMD_SEARCH_LISTS=0" "SRRG2_AMD_SEARCH_LISTS=2" "SRRG2_AMD_SEARCH_LISTS=0" "SRRG2_AMD_SEARCH_LISTS=2"
done
cat $O/ab_lists.txt
cd /tmp
B="--workload c4 --batch 256 --steps 1 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace -d /tmp/tr256 -o t -- python $R/bench.py --workload c4 --batch 256 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
echo "durations" > $O/c4_256_passes_cnl.txt; python $R/tools/iter_durations.py $(find /tmp/tr256 -name '*.db' | head -1) 10 >> $O/c4_256_passes_cnl.txt
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d /tmp/p256a -o p -- python $R/bench.py $B > /dev/null 2>&1
python $R/tools/iter_durations.py $(find /tmp/p256a -name '*.db' | head -1) 10 >> $O/c4_256_passes_cnl.txt
timeout 600 rocprofv3 --pmc TA_BUSY_avr SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d /tmp/p256b -o p -- python $R/bench.py $B > /dev/null 2>&1
python $R/tools/iter_durations.py $(find /tmp/p256b -name '*.db' | head -1) 10 >> $O/c4_256_passes_cnl.txt
cat $O/c4_256_passes_cnl.txt
