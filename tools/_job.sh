cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3c; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
for K in 32 256; do
SRRG2_AMD_LIB=$R/srrg2_slam_interfaces_amd/lib/libsrrg2_slam_amd_stats.so SRRG2_AMD_LDS_TILE=1 python tools/tile_stats.py $K > $O/tile_stats_$K.txt 2>&1
done
SRRG2_AMD_LIB=$R/srrg2_slam_interfaces_amd/lib/libsrrg2_slam_amd_stats.so SRRG2_AMD_LDS_TILE=1 SRRG2_AMD_MSORT_BITS=-1 python tools/tile_stats.py 256 > $O/tile_stats_256_isosort.txt 2>&1
cat $O/tile_stats_*.txt
cd /tmp
rocprofv3 -L > $O/counters_avail.txt 2>&1
for t in 0 1; do
  SRRG2_AMD_LDS_TILE=$t timeout 300 rocprofv3 --kernel-trace -d /tmp/tr_$t -o t -- python $R/bench.py --workload c4 --batch 32 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  echo "tile=$t" >> $O/iter_durations.txt; python $R/tools/iter_durations.py $(find /tmp/tr_$t -name '*.db' | head -1) 10 >> $O/iter_durations.txt
  SRRG2_AMD_LDS_TILE=$t timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d /tmp/p1_$t -o p -- python $R/bench.py --workload c4 --batch 32 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python $R/tools/iter_durations.py $(find /tmp/p1_$t -name '*.db' | head -1) 10 >> $O/iter_durations.txt
  SRRG2_AMD_LDS_TILE=$t timeout 300 rocprofv3 --pmc TA_BUSY_avr SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d /tmp/p2_$t -o p -- python $R/bench.py --workload c4 --batch 32 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python $R/tools/iter_durations.py $(find /tmp/p2_$t -name '*.db' | head -1) 10 >> $O/iter_durations.txt
  SRRG2_AMD_LDS_TILE=$t timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU -d /tmp/p3_$t -o p -- python $R/bench.py --workload c4 --batch 32 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python $R/tools/iter_durations.py $(find /tmp/p3_$t -name '*.db' | head -1) 10 >> $O/iter_durations.txt
done
cat $O/iter_durations.txt
