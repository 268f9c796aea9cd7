#!/bin/bash
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu --timeout 600 --durations=5 2>&1 | tail -25
