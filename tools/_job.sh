#!/bin/bash
# scratch: the command file of the last `gpurun -- 'bash tools/_job.sh'` call of the session
cd /root/repo
mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/r1l_bench_c2.json 2>/dev/null
timeout 300 python bench.py --workload c3 --no-cpu-all-cores --no-cpu-baseline > gpurun_out/r1l_bench_c3.json 2>/dev/null
timeout 300 python bench.py --workload c4 --no-cpu-all-cores --no-cpu-baseline > gpurun_out/r1l_bench_c4.json 2>/dev/null
timeout 300 python bench.py --workload c4 --batch 256 --steps 20 --warmup 3 --no-cpu-all-cores --no-cpu-baseline > gpurun_out/r1l_bench_c4_256.json 2>/dev/null
for f in c2 c3 c4 c4_256; do python -c "
import json
d=json.loads(open('gpurun_out/r1l_bench_$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['frac'],4), d['roofline']['avg_launch_ms'], d.get('speedup_vs_cpu_baseline'), d.get('speedup_vs_cpu_all_cores'))"; done
