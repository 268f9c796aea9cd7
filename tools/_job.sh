cd $GRAFT_REPO_ROOT
for c in 1000.0f 0.5f 0.25f 0.1f; do
make -B -C srrg2_slam_interfaces_amd/csrc EXTRA=-DPAD_CAP=$c > /dev/null 2>&1
echo "cap $c c2"; python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
echo "cap $c c4"; python bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
