cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
echo "c2"; python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
echo "c2 ov 0.7"; python bench.py --overlap 0.7 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
echo "c4"; python bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_c2i -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_c4i -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_steps.py $(ls gpurun_out/prof_c2i/*/*.db | head -1)
python tools/trace_steps.py $(ls gpurun_out/prof_c4i/*/*.db | head -1)
