#!/bin/bash
# scratch: the command file of the last `gpurun -- 'bash tools/_job.sh'` call of the session (GPU tests, smoke, bench lines, traces)
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 300 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/r1k_bench_c2.json 2>/dev/null
timeout 300 python bench.py --workload c3 --no-cpu-all-cores --no-cpu-baseline > gpurun_out/r1k_bench_c3.json 2>/dev/null
timeout 300 python bench.py --workload c4 --no-cpu-all-cores --no-cpu-baseline > gpurun_out/r1k_bench_c4.json 2>/dev/null
timeout 300 python bench.py --workload c4 --batch 256 --steps 20 --warmup 3 --no-cpu-all-cores --no-cpu-baseline > gpurun_out/r1k_bench_c4_256.json 2>/dev/null
timeout 600 python tools/bench_posegraph.py > gpurun_out/r1k_bench_c5.json 2>/dev/null
timeout 300 python tools/bench_tracker.py > gpurun_out/r1k_bench_tracker.json 2>/dev/null
timeout 300 python tools/bench_small.py --beams 360 2000 4000 > gpurun_out/r1k_bench_small.json 2>/dev/null
for w in c2 c3 c4; do
  rm -rf /tmp/tr_$w
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tr_$w -o t -- python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-all-cores --no-cpu-baseline > /tmp/tr_$w.log 2>&1
  python tools/rocpd_summary.py gpurun_out/r1k_rocprofv3_${w}_summary.txt kernel_trace_stats=$(ls /tmp/tr_$w/*.db | head -1)
done
python tools/trace_steps.py $(ls /tmp/tr_c2/*.db | head -1) | cut -c1-300
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p -- python bench.py --steps 20 --warmup 2 --no-cpu-all-cores --no-cpu-baseline > /tmp/pmc_$c.log 2>&1
  cp $(ls /tmp/pmc_$c/*.db | head -1) /tmp/pmc_$c.db
done
python tools/traffic_from_pmc.py gpurun_out/r1k_traffic_c2.json c2 /tmp/pmc_FETCH_SIZE.db /tmp/pmc_WRITE_SIZE.db | grep bytes_per
for f in c2 c3 c4 c4_256; do python -c "
import json
d=json.loads(open('gpurun_out/r1k_bench_$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['frac'],4), d['roofline']['avg_launch_ms'], d.get('speedup_vs_cpu_baseline'), d.get('speedup_vs_cpu_all_cores'))"; done
tail -c 250 gpurun_out/r1k_bench_c5.json; echo; cut -c1-330 gpurun_out/r1k_bench_tracker.json
