cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_r1_j.json 2> gpurun_out/bench_r1_j.err; tail -1 gpurun_out/bench_r1_j.json | cut -c1-160
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats -d $R/r1g_c2_trace -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1; echo trace rc=$?
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/r1g_c2_pmc_$set -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1; echo "$set rc=$?"
done
cd $GRAFT_REPO_ROOT
python tools/bench_tracker.py 2>&1 | tail -1 > gpurun_out/bench_r1_j_tracker.json
