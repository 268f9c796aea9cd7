cd $GRAFT_REPO_ROOT
for c in 4 6 8 12 16 24; do
echo "target $c c2"; SRRG2_AMD_CELL_TARGET=$c python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
echo "target $c c4"; SRRG2_AMD_CELL_TARGET=$c python bench.py --workload c4 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
