#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$R/gpurun_out/r4r; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sharing or c3 or projective" > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
python bench.py --workload c3 --steps 100 --warmup 10 --no-cpu-baseline --c3-own-clouds | cut -c1-160
python bench.py --workload c3 --steps 100 --warmup 10 --no-cpu-baseline | cut -c1-160
python bench.py --workload c3 --steps 100 --warmup 10 --no-cpu-baseline --c3-own-clouds | cut -c1-160
python bench.py --workload c3 --steps 100 --warmup 10 --no-cpu-baseline | cut -c1-160
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d /tmp/pmc_c3_$c -o p -- python $R/bench.py --workload c3 --steps 205 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
python $R/tools/traffic_from_pmc.py $O/traffic_c3.json c3 $(find /tmp/pmc_c3_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/pmc_c3_WRITE_SIZE -name '*.db' | head -1) | head -30
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_c3 -o t -- python $R/bench.py --workload c3 --steps 205 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py $O/rocprofv3_c3_summary.txt kernel_trace_stats=$(find /tmp/tr_c3 -name '*.db' | head -1); head -8 $O/rocprofv3_c3_summary.txt | cut -c1-140
