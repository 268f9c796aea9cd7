cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2zl; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_posegraph.py -m gpu -x -q > $O/pytest_pg.txt 2>&1; tail -15 $O/pytest_pg.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
SRRG2_AMD_PG_DEBUG=1 timeout 600 python tools/bench_posegraph.py 2>$O/c5.err | cut -c100-230; tail -1 $O/c5.err | cut -c1-400
