set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2k; mkdir -p $O
(time timeout 1500 python -m pytest tests/test_gpu_posegraph.py tests/test_gpu_graph_lifecycle.py tests/test_gpu_full_size.py tests/test_cpp_mirror.py -q -m gpu 2>&1 | tail -12) > $O/pytest.log 2>&1
timeout 600 python tools/bench_posegraph.py 50000 200000 --cpu > $O/bench_c5.json 2> $O/bench_c5.err
