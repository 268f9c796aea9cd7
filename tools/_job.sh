#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r3s
O=gpurun_out/r3s
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tuning.py tests/test_golden.py tests/test_gpu_full_size.py -m gpu -x -q > $O/pytest_odd_order.log 2>&1
tail -2 $O/pytest_odd_order.log
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
