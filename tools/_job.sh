cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_scene.py -x -q 2>&1 | tail -3
python tools/bench_scene.py 2>&1 | tail -1
python tools/bench_scene.py --points 20000000 2>&1 | tail -1 > gpurun_out/bench_scene.json; cat gpurun_out/bench_scene.json
