#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 120 > gpurun_out/p_parity.log 2>&1
tail -3 gpurun_out/p_parity.log
SRRG2_AMD_TUNE=2097152 SRRG2_AMD_PERSIST_DBG=gpurun_out/p_stamps.bin timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-all-cores --no-cpu-baseline > gpurun_out/p_bench_dbg.json 2> gpurun_out/p_bench_dbg.err
python tools/persist_stamps.py gpurun_out/p_stamps.bin
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-all-cores --no-cpu-baseline > gpurun_out/p_bench_c2.json 2> gpurun_out/p_bench_c2.err
python - <<'PY'
import json
for f in ("p_bench_c2",):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["config"])
    except Exception as e: print(f, "ERR", e)
PY
