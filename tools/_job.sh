set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2j; mkdir -p $O
(time timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15) > $O/pytest.log 2>&1
