cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2zt; mkdir -p $O
for n in 100000 200000 500000 1000000; do for t in 0 512; do
echo "points $n tune $t $(SRRG2_AMD_TUNE=$t python bench.py --workload c2 --points $n --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c40-160)"
done; done | tee $O/queue_vs_inkernel.txt
for ov in 0.6 0.3; do for t in 0 512; do
echo "overlap $ov tune $t $(SRRG2_AMD_TUNE=$t python bench.py --workload c2 --overlap $ov --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c40-160)"
done; done | tee -a $O/queue_vs_inkernel.txt
python tools/bench_tracker.py 2>/dev/null | cut -c1-400 | tee -a $O/queue_vs_inkernel.txt
SRRG2_AMD_TUNE=512 python tools/bench_tracker.py 2>/dev/null | cut -c1-400 | tee -a $O/queue_vs_inkernel.txt
echo lag; SRRG2_AMD_PG_DEBUG=0 python tools/bench_posegraph.py 2>/dev/null | cut -c100-330 | tee $O/c5_lag.txt
SRRG2_AMD_PG_LAG=0 python tools/bench_posegraph.py 2>/dev/null | cut -c100-330 | tee -a $O/c5_lag.txt
timeout 600 python -m pytest tests/test_gpu_posegraph.py tests/test_gpu_graph_lifecycle.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -2
