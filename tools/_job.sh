#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$R/gpurun_out/r4m; mkdir -p $O
timeout 1500 python -m pytest tests/test_golden.py tests/test_gpu_posegraph.py tests/test_multi_gpu_gloo.py tests/test_cpp_mirror.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
