cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for q in -1 1; do
echo "QPROBE $q: full / ov0.7 / 10k"
for extra in "" "--overlap 0.7" "--points 10000"; do
SRRG2_AMD_QPROBE=$q python bench.py $extra --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('   ', round(d['value']), d['ms_per_step'])"
done
done
python tools/bench_tracker.py 2>&1 | tail -1 | cut -c60-330
