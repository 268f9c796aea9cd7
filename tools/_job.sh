cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2zz4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_posegraph.py tests/test_gpu_graph_lifecycle.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -2
for r in 1 2; do python tools/bench_posegraph.py 2>/dev/null | cut -c100-230; done
cd /tmp
SRRG2_AMD_PG_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_c5 -o t -- python $GRAFT_REPO_ROOT/tools/bench_posegraph.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $O/rocprofv3_c5_summary.txt kernel_trace_stats=$(find /tmp/tr_c5 -name '*.db' | head -1)
grep "coarsest_inverse" $O/rocprofv3_c5_summary.txt | cut -c1-200
