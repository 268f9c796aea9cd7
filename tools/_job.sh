#!/bin/bash
cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 300 2>&1 | tail -3
python tools/loop_compute.py 100000 200
python tools/loop_compute.py 10000 200
timeout 300 python tools/bench_tracker.py 2>/dev/null | tail -1 | cut -c1-420
