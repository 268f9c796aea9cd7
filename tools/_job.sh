# scratch job for `gpurun -- 'bash tools/_job.sh'` (edited per experiment); the committed default runs the GPU suite
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3
