cd $GRAFT_REPO_ROOT
timeout 300 tests/cpp/bin/test_loop_closure 2>&1 | tail -12
timeout 600 python -m pytest tests/test_cpp_mirror.py tests/test_multi_gpu_gloo.py -m gpu -x -q 2>&1 | tail -3
