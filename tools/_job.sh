cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "partial or trimmed" 2>&1 | tail -5
