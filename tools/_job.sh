cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_given_correspondences.py -x -q -m gpu 2>&1 | tail -2
python tools/bench_hbst.py --cpu 2>&1 | tail -1 | tee gpurun_out/bench_hbst.json
