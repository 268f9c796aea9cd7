cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2zn; mkdir -p $O
SRRG2_AMD_GRID2=2 timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_grid2.txt 2>&1; tail -3 $O/pytest_grid2.txt
for rep in 1 2; do for s in 0 2 3; do
  echo "grid2 $s c2 $(SRRG2_AMD_GRID2=$s python bench.py --workload c2 --no-cpu-baseline 2>/dev/null | cut -c40-160)"
  echo "grid2 $s c4 $(SRRG2_AMD_GRID2=$s python bench.py --workload c4 --no-cpu-baseline 2>/dev/null | cut -c40-160)"
  echo "grid2 $s c4-256 $(SRRG2_AMD_GRID2=$s python bench.py --workload c4 --batch 256 --no-cpu-baseline 2>/dev/null | cut -c40-160)"
done; done | tee $O/ab_grid2.txt
cd /tmp
for s in 0 2; do
SRRG2_AMD_GRID2=$s timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_s$s -o t -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
SRRG2_AMD_GRID2=$s timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr2_s$s -o t -- python $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python - $s <<'PY'
import sqlite3, glob, os, sys
t=sys.argv[1]
out = open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r2zn/iter_durations.txt', 'a')
for tag, d in (('c4', '/tmp/tr_s%s'%t), ('c2', '/tmp/tr2_s%s'%t)):
    cur2 = sqlite3.connect(glob.glob(d+'/**/*.db', recursive=True)[0]).cursor()
    dur = [(r[0], r[1]/1000) for r in cur2.execute("select name, (end-start) from kernels where name like '%k_icp_step%' order by start")]
    out.write("grid2 %s %s: %s\n" % (t, tag, [(n.split('<')[0][-10:], round(x,1)) for n, x in dur[-13:]]))
PY
done
cat $O/iter_durations.txt
