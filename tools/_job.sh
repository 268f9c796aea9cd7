cd $GRAFT_REPO_ROOT
make -B -C srrg2_slam_interfaces_amd/csrc EXTRA="-DSRRG2_TIMELINE" > /dev/null 2>&1
SRRG2_AMD_TIMELINE=$GRAFT_REPO_ROOT/gpurun_out/tl.bin python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-100
python tools/timeline.py gpurun_out/tl.bin 0 1 2
