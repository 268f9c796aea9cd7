#!/bin/bash
# scratch job for gpurun
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t35
python -m pytest tests -q -m gpu 2>&1 | tail -2 > gpurun_out/t35/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/t35/smoke.txt 2>&1
