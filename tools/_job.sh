#!/bin/bash
# final profile refresh (r1i): C2 bench line with CPU baselines + rocprofv3 summary + PMC traffic; C4 bench lines
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 300 > gpurun_out/t_gpu.log 2>&1
tail -2 gpurun_out/t_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/r1i_bench_c2.json 2> gpurun_out/r1i_bench_c2.err
timeout 300 python bench.py --workload c4 --no-cpu-all-cores --no-cpu-baseline > gpurun_out/r1i_bench_c4.json 2>/dev/null
timeout 300 python bench.py --workload c4 --batch 256 --steps 20 --warmup 3 --no-cpu-all-cores --no-cpu-baseline > gpurun_out/r1i_bench_c4_256.json 2>/dev/null
timeout 300 python bench.py --workload c3 --no-cpu-all-cores --no-cpu-baseline > gpurun_out/r1i_bench_c3.json 2>/dev/null
rm -rf /tmp/tr_c2
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tr_c2 -o t -- python bench.py --steps 100 --warmup 10 --no-cpu-all-cores --no-cpu-baseline > /tmp/tr_c2.log 2>&1
python tools/rocpd_summary.py gpurun_out/r1i_rocprofv3_c2_summary.txt kernel_trace_stats=$(ls /tmp/tr_c2/*.db | head -1)
python tools/trace_steps.py $(ls /tmp/tr_c2/*.db | head -1) | cut -c1-300
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p -- python bench.py --steps 20 --warmup 2 --no-cpu-all-cores --no-cpu-baseline > /tmp/pmc_$c.log 2>&1
  cp $(ls /tmp/pmc_$c/*.db | head -1) /tmp/pmc_$c.db
done
python tools/traffic_from_pmc.py gpurun_out/r1i_traffic_c2.json c2 /tmp/pmc_FETCH_SIZE.db /tmp/pmc_WRITE_SIZE.db | grep bytes_per
for f in c2 c3 c4 c4_256; do python -c "
import json
d=json.loads(open('gpurun_out/r1i_bench_$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['frac'],4), d['roofline']['avg_launch_ms'], d.get('speedup_vs_cpu_baseline'), d.get('speedup_vs_cpu_all_cores'))"; done
