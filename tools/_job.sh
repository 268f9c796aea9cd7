cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_full_size.py -m gpu -x -q -k "sparse_direct" 2>&1 | tail -8
