cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2zk; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for rep in 1 2; do for g in 1 2 4 8; do
  echo "segments $g c4-32 $(SRRG2_AMD_MSORT_SEGMENTS=$g python bench.py --workload c4 --no-cpu-baseline 2>/dev/null | cut -c40-160)"
done; done | tee $O/ab_msort_segments.txt
for g in 1 4; do echo "segments $g c4-8 $(SRRG2_AMD_MSORT_SEGMENTS=$g python bench.py --workload c4 --batch 8 --no-cpu-baseline 2>/dev/null | cut -c40-160)"; done | tee -a $O/ab_msort_segments.txt
for g in 1 2; do echo "segments $g c4-64 $(SRRG2_AMD_MSORT_SEGMENTS=$g python bench.py --workload c4 --batch 64 --no-cpu-baseline 2>/dev/null | cut -c40-160)"; done | tee -a $O/ab_msort_segments.txt
cd /tmp
for g in 1 4; do
SRRG2_AMD_MSORT_SEGMENTS=$g timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_g$g -o t -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $O/rocprofv3_c4_segments$g.txt kernel_trace_stats=$(find /tmp/tr_g$g -name '*.db' | head -1)
head -6 $O/rocprofv3_c4_segments$g.txt | cut -c1-140
done
