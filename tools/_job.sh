#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t23
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/t23/pytest_gpu.txt
