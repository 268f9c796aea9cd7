set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2v; mkdir -p $O
cp srrg2_slam_interfaces_amd/lib/libsrrg2_knobs.so srrg2_slam_interfaces_amd/lib/libsrrg2_slam_amd.so
cd /tmp
for t in 0 4194304 8388608 16777216 33554432; do
SRRG2_AMD_TUNE=$t timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_$t -o t -- python $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python - $t <<'PY'
import sqlite3, glob, os, sys
t=sys.argv[1]
db2 = glob.glob('/tmp/tr_%s/**/*.db'%t, recursive=True)[0]
cur2 = sqlite3.connect(db2).cursor()
out = open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r2v/control_knobs.txt', 'a')
for pat in ('%k_icp_control(%',):
    dur = sorted([r[0]/1000 for r in cur2.execute("select (end-start) from kernels where name like ? order by start", (pat,))])
    out.write("tune %s %s n %d avg %.2f median %.2f min %.2f\n" % (t, pat, len(dur), sum(dur)/len(dur), dur[len(dur)//2], dur[0]))
PY
done
cat $O/control_knobs.txt
