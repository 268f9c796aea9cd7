#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$R/gpurun_out/r4s; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d /tmp/tr32 -o t -- python $R/bench.py --workload c4 --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/batch_timeline.py $(find /tmp/tr32 -name '*.db' | head -1) > $O/timeline_c4_32.txt
tail -60 $O/timeline_c4_32.txt
SRRG2_AMD_BATCH_PIPELINE=0 timeout 600 rocprofv3 --kernel-trace -d /tmp/tr32b -o t -- python $R/bench.py --workload c4 --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/batch_timeline.py $(find /tmp/tr32b -name '*.db' | head -1) > $O/timeline_c4_32_unsplit.txt
tail -3 $O/timeline_c4_32_unsplit.txt
