#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 300 python tools/bench_small.py --beams 360 2000 4000 > gpurun_out/r1i_bench_small.json 2>/dev/null; python -c "
import json
for c in json.load(open('gpurun_out/r1i_bench_small.json'))['small_alignments']: print('  %-40s gpu %.4f ms oracle %.3f  x%.1f identical %s' % (c['case'], c['gpu_ms'], c['oracle_ms'], c['speedup'], c['X_bit_identical']))"
timeout 300 python tools/bench_tracker.py > gpurun_out/r1i_bench_tracker.json 2>/dev/null
