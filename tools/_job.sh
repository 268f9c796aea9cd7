cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
echo c2; python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('   ', round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_ms'])"
echo c3; python bench.py --workload c3 --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('   ', round(d['value']), d['ms_per_step'])"
echo c4; python bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('   ', round(d['value']), d['ms_per_step'])"
python tools/bench_tracker.py 2>&1 | tail -1 | cut -c1-420
