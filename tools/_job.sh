#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t26
python -m pytest tests/test_gpu_graph_lifecycle.py tests/test_gpu_posegraph.py -x -q 2>&1 | tail -15 > gpurun_out/t26/pytest.txt
python bench.py --workload c5 > gpurun_out/t26/bench_c5.json 2>gpurun_out/t26/bench_c5.err
