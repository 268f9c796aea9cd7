cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2zi; mkdir -p $O
L=srrg2_slam_interfaces_amd/lib
cp $L/libsrrg2_slam_amd.so /tmp/new.so; cp $L/libsrrg2_old.so /tmp/old.so
for rep in 1 2; do for v in old new; do
  cp /tmp/$v.so $L/libsrrg2_slam_amd.so
  echo "$v c2 $(python bench.py --workload c2 --no-cpu-baseline 2>/dev/null | cut -c40-160)"
  echo "$v c4 $(python bench.py --workload c4 --no-cpu-baseline 2>/dev/null | cut -c40-160)"
  echo "$v c4-256 $(python bench.py --workload c4 --batch 256 --no-cpu-baseline 2>/dev/null | cut -c40-160)"
done; done | tee $O/ab_nowait.txt
cp /tmp/new.so $L/libsrrg2_slam_amd.so
