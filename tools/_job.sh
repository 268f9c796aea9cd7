cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3n; mkdir -p $O; R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
export SRRG2_AMD_LIB=$R/srrg2_slam_interfaces_amd/lib/libsrrg2_slam_amd_knobs.so
rm -f $O/knob_attribution_tile.txt
for tune in 0 3 67108867 33554435 16; do
  rm -rf /tmp/p1
  SRRG2_AMD_TUNE=$tune timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d /tmp/p1 -o p -- python $R/bench.py --workload c4 --batch 32 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  echo "tune=$tune" >> $O/knob_attribution_tile.txt
  python $R/tools/iter_durations.py $(find /tmp/p1 -name '*.db' | head -1) 10 | cut -c1-75 >> $O/knob_attribution_tile.txt
done
cat $O/knob_attribution_tile.txt
