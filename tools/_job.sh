cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for n in 100000 300000 1000000; do
echo "c2 $n points"; timeout 600 python bench.py --points $n --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
