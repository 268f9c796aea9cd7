cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2zj; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for rep in 1 2; do for l in 1 2; do
  echo "lpq $l c2 $(SRRG2_AMD_LPQ=$l python bench.py --workload c2 --no-cpu-baseline 2>/dev/null | cut -c40-160)"
done; done | tee $O/ab_lpq.txt
echo "lpq 2 c2 200k: $(SRRG2_AMD_LPQ=2 python bench.py --workload c2 --points 200000 --no-cpu-baseline 2>/dev/null | cut -c40-160)"
echo "lpq 1 c2 200k: $(SRRG2_AMD_LPQ=1 python bench.py --workload c2 --points 200000 --no-cpu-baseline 2>/dev/null | cut -c40-160)"
cd /tmp
for l in 1 2; do
SRRG2_AMD_LPQ=$l timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_l$l -o t -- python $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $O/rocprofv3_c2_lpq$l.txt kernel_trace_stats=$(find /tmp/tr_l$l -name '*.db' | head -1)
head -8 $O/rocprofv3_c2_lpq$l.txt | cut -c1-140
done
