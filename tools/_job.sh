cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2zz2; mkdir -p $O
for n in 100000 200000 500000 1000000 2000000 5000000; do for g in 0 -1; do
  if [ $g = 0 ]; then export SRRG2_AMD_GRID2=0; else unset SRRG2_AMD_GRID2; fi
  echo "points $n grid2 $g $(python bench.py --workload c2 --points $n --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c40-160)"
done; done | tee $O/grid2_dense.txt
unset SRRG2_AMD_GRID2
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_big -o t -- python $GRAFT_REPO_ROOT/bench.py --workload c2 --points 2000000 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/trace_steps.py $(find /tmp/tr_big -name '*.db' | head -1) | cut -c1-400 | tee $O/trace_2M.txt
