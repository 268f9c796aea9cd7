#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 300 -k "batch" 2>&1 | tail -15
