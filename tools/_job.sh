cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2zw; mkdir -p $O
timeout 900 python -m pytest tests/test_multi_gpu_gloo.py -m gpu -x -q -k rccl > $O/pytest_mg.txt 2>&1; tail -30 $O/pytest_mg.txt
