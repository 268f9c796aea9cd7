cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -3
python bench.py > gpurun_out/bench_r1_g.json 2> gpurun_out/bench_r1_g.err; tail -1 gpurun_out/bench_r1_g.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats -d $R/r1d_c2_trace -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1; echo trace rc=$?
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/r1d_c2_pmc_$n -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1; echo "$n rc=$?"
done
timeout 300 rocprofv3 --kernel-trace --stats -d $R/r1d_c4_trace -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; echo c4 rc=$?
timeout 300 rocprofv3 --kernel-trace --stats -d $R/r1d_c3_trace -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2>&1; echo c3 rc=$?
cd $GRAFT_REPO_ROOT
python bench.py --workload c3 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_r1_g_c3.json
python bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_r1_g_c4.json
python tools/bench_posegraph.py 2>&1 | tail -3
