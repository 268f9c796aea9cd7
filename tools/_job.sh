#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 300 > gpurun_out/t_gpu.log 2>&1
tail -2 gpurun_out/t_gpu.log
for w in c2 c3 c4; do
timeout 300 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-all-cores --no-cpu-baseline > gpurun_out/q_bench_$w.json 2>/dev/null
done
timeout 300 python tools/bench_tracker.py > gpurun_out/q_tracker.json 2>&1
python - <<'PY'
import json
for f in ("q_bench_c2","q_bench_c3","q_bench_c4"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"])
    except Exception as e: print(f, "ERR", e)
PY
tail -1 gpurun_out/q_tracker.json | cut -c1-400
