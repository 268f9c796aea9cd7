# scratch job for `gpurun -- 'bash tools/_job.sh'` (edited per experiment)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_given_correspondences.py tests/test_loop_detector.py -m gpu -q 2>&1 | tail -2
python tools/bench_hbst.py 2>/dev/null | cut -c1-300
