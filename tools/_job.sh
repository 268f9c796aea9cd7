set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2l; mkdir -p $O
(timeout 600 tests/cpp/bin/test_loop_closure) > $O/loop.log 2>&1
(time timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -8) > $O/pytest.log 2>&1
