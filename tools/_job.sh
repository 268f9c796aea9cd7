#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$R/gpurun_out/r4q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tuning.py -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
bash tools/ab_env.sh $O/ab.txt "--workload c4 --batch 256 --steps 10 --warmup 2" "-" "-"
bash tools/ab_env.sh $O/ab.txt "--workload c4 --batch 32 --steps 30 --warmup 3" "-" "-"
bash tools/ab_env.sh $O/ab.txt "--workload c4 --batch 8 --steps 30 --warmup 3" "-"
bash tools/ab_env.sh $O/ab.txt "--workload c2 --steps 200 --warmup 20" "-" "-"
cat $O/ab.txt
cd /tmp
B="--workload c4 --batch 256 --steps 1 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d /tmp/p256a -o p -- python $R/bench.py $B > /dev/null 2>&1
python $R/tools/iter_durations.py $(find /tmp/p256a -name '*.db' | head -1) 10 > $O/c4_256_passes.txt
SRRG2_AMD_BATCH_PIPELINE=0 timeout 600 rocprofv3 --kernel-trace -d /tmp/tr256 -o t -- python $R/bench.py --workload c4 --batch 256 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/iter_durations.py $(find /tmp/tr256 -name '*.db' | head -1) 10 >> $O/c4_256_passes.txt
cat $O/c4_256_passes.txt
