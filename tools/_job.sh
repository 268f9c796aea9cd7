#!/bin/bash
# scratch job for gpurun
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t32
python -m pytest tests -q -m gpu 2>&1 | tail -3 > gpurun_out/t32/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/t32/smoke.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/t32/bench_driver.json 2> gpurun_out/t32/bench_driver.err
