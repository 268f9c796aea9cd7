set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2c
(time timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5) > gpurun_out/r2c/pytest.log 2>&1
B='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"])'
for q in 0 1; do for ppt in 1 2 4; do
echo "== FAST_QUEUE=$q PPT=$ppt c4" >> gpurun_out/r2c/sweep.log
SRRG2_AMD_FAST_QUEUE=$q SRRG2_AMD_FAST_PPT=$ppt timeout 300 python bench.py --workload c4 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "$B" >> gpurun_out/r2c/sweep.log 2>&1
done; done
for ppt in 1 2; do
echo "== FAST_QUEUE=0 PPT=$ppt c4 256" >> gpurun_out/r2c/sweep.log
SRRG2_AMD_FAST_QUEUE=0 SRRG2_AMD_FAST_PPT=$ppt timeout 300 python bench.py --workload c4 --batch 256 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "$B" >> gpurun_out/r2c/sweep.log 2>&1
done
for ff in 3 4; do
echo "== FAST_FROM=$ff c2" >> gpurun_out/r2c/sweep.log
SRRG2_AMD_FAST_FROM=$ff timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "$B" >> gpurun_out/r2c/sweep.log 2>&1
done
cd /tmp
SRRG2_AMD_FAST_QUEUE=0 SRRG2_AMD_FAST_PPT=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_c4 -o t -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/r2c/trace_c4_summary.txt trace=$(find /tmp/tr_c4 -name "*.db" | head -1)
SRRG2_AMD_FAST_QUEUE=0 SRRG2_AMD_FAST_PPT=2 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_c4b -o t -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/r2c/trace_c4b_summary.txt trace=$(find /tmp/tr_c4b -name "*.db" | head -1)
