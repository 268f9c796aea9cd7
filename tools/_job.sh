cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_graph_lifecycle.py tests/test_gpu_posegraph.py -x -q 2>&1 | tail -3
python tools/bench_posegraph.py 2>&1 | tail -1
