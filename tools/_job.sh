cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2zr; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for r in 1 2; do python bench.py --workload c3 --no-cpu-baseline 2>/dev/null | cut -c1-170; python bench.py --workload c2 --no-cpu-baseline 2>/dev/null | cut -c1-170; done | tee $O/bench.txt
cd /tmp
for w in c3 c2; do
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_$w -o t -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $O/rocprofv3_${w}_summary.txt kernel_trace_stats=$(find /tmp/tr_$w -name '*.db' | head -1)
grep "k_icp_control" $O/rocprofv3_${w}_summary.txt | cut -c1-140
done
