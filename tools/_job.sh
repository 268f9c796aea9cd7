#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 300 > gpurun_out/t_gpu.log 2>&1
tail -3 gpurun_out/t_gpu.log
timeout 300 python bench.py --workload c4 --steps 100 --warmup 10 --no-cpu-all-cores --no-cpu-baseline > gpurun_out/q_bench_c4.json 2> gpurun_out/q_bench_c4.err
timeout 300 python bench.py --workload c4 --batch 256 --steps 20 --warmup 3 --no-cpu-all-cores --no-cpu-baseline > gpurun_out/q_bench_c4_256.json 2> gpurun_out/q_bench_c4_256.err
python - <<'PY'
import json
for f in ("q_bench_c4","q_bench_c4_256"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"])
    except Exception as e: print(f, "ERR", e)
PY
