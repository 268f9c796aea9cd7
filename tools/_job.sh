#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 300 > gpurun_out/t_gpu.log 2>&1
tail -2 gpurun_out/t_gpu.log
timeout 300 python tools/bench_small.py --beams 360 2000 4000 > gpurun_out/r1h_bench_small.json 2>/dev/null; python -c "
import json
for c in json.load(open('gpurun_out/r1h_bench_small.json'))['small_alignments']: print('  %-40s gpu %.4f ms oracle %.3f  identical %s' % (c['case'], c['gpu_ms'], c['oracle_ms'], c['X_bit_identical']))"
