cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for t in 0 131072; do
echo "tune $t c3"; SRRG2_AMD_TUNE=$t python bench.py --workload c3 --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done
