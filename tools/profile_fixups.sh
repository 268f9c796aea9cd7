#!/bin/bash
# The three artefacts of tools/profile_round.sh that name kernels (C2's PMC traffic, the pipelined C4-32 timeline, C2's kernel timeline), alone.
# usage: gpurun -- 'bash tools/profile_fixups.sh TAG'
TAG=${1:-round}
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$(pwd); export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d /tmp/pmc_c2_$c -o p -- python $R/bench.py --workload c2 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
python $R/tools/traffic_from_pmc.py $O/traffic_c2.json c2 $(find /tmp/pmc_c2_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/pmc_c2_WRITE_SIZE -name '*.db' | head -1)
timeout 600 rocprofv3 --kernel-trace -d /tmp/tr32 -o t -- python $R/bench.py --workload c4 --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/batch_timeline.py $(find /tmp/tr32 -name '*.db' | head -1) > $O/timeline_c4_32_pipelined.txt
timeout 600 rocprofv3 --kernel-trace -d /tmp/tr_c2 -o t -- python $R/bench.py --workload c2 --steps 100 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/compute_span.py $(find /tmp/tr_c2 -name '*.db' | head -1) > $O/c2_compute_span.txt
SRRG2_AMD_TUNE=8388608 timeout 600 rocprofv3 --kernel-trace -d /tmp/tr_c2b -o t -- python $R/bench.py --workload c2 --steps 100 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
echo "# SRRG2_AMD_TUNE=8388608 (bit 23: the k_icp_init launch in front of every compute(), as before)" >> $O/c2_compute_span.txt
python $R/tools/compute_span.py $(find /tmp/tr_c2b -name '*.db' | head -1) >> $O/c2_compute_span.txt
SRRG2_AMD_TUNE=41943040 timeout 600 rocprofv3 --kernel-trace -d /tmp/tr_c2c -o t -- python $R/bench.py --workload c2 --steps 100 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
echo "# SRRG2_AMD_TUNE=41943040 (bits 23 + 25: k_icp_init launch AND the 256-thread k_icp_control_final: the start of the session)" >> $O/c2_compute_span.txt
python $R/tools/compute_span.py $(find /tmp/tr_c2c -name '*.db' | head -1) >> $O/c2_compute_span.txt
cat $O/c2_compute_span.txt | grep -E "steady|SRRG2"; head -30 $O/timeline_c4_32_pipelined.txt; python -c "import json; d=json.load(open('$O/traffic_c2.json')); print(d['bytes_per_slice_pass'], list(d['kernels']))"
