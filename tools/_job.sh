cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/bench_r1_i.json 2> gpurun_out/bench_r1_i.err; tail -1 gpurun_out/bench_r1_i.json | cut -c1-160
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats -d $R/r1f_c2_trace -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1; echo trace rc=$?
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/r1f_c2_pmc_$set -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1; echo "$set rc=$?"
done
timeout 300 rocprofv3 --kernel-trace --stats -d $R/r1f_c4_trace -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; echo c4 rc=$?
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/r1f_c4_pmc_FETCH_SIZE -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; echo c4f rc=$?
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/r1f_c4_pmc_WRITE_SIZE -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; echo c4w rc=$?
timeout 300 rocprofv3 --kernel-trace --stats -d $R/r1f_c3_trace -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2>&1; echo c3 rc=$?
timeout 300 rocprofv3 --kernel-trace --stats -d $R/r1f_c5_trace -- python $GRAFT_REPO_ROOT/tools/bench_posegraph.py > /dev/null 2>&1; echo c5 rc=$?
cd $GRAFT_REPO_ROOT
python bench.py --workload c3 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_r1_i_c3.json
python bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_r1_i_c4.json
python bench.py --workload c4 --batch 256 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_r1_i_c4_256.json
python tools/bench_posegraph.py 2>&1 | tail -1 > gpurun_out/bench_r1_i_c5.json
python tools/bench_scene.py --points 20000000 2>&1 | tail -1 > gpurun_out/bench_r1_i_scene.json
