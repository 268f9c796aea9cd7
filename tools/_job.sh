cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
python tools/bench_tracker.py 2>&1 | tail -1 | tee gpurun_out/bench_tracker.json
