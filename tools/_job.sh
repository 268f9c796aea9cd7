cd $GRAFT_REPO_ROOT
for p in 2 3; do
echo "PPT=$p"; SRRG2_AMD_PPT=$p timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batch" 2>&1 | tail -2
done
for p in 1 2 4; do
echo "ppt $p c4"; SRRG2_AMD_PPT=$p python bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
echo "c2"; python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
