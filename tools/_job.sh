#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
for t in 0; do
rm -rf /tmp/k_$t
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM -d /tmp/k_$t -o k -- python bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-all-cores --no-cpu-baseline > /tmp/k_$t.log 2>&1
python - <<PY
import sqlite3,glob,collections
db=glob.glob("/tmp/k_$t/*.db")[0]
cur=sqlite3.connect(db).cursor()
rows=list(cur.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection where kernel_name like '%k_icp_step<%' order by dispatch_id"))
d=collections.defaultdict(dict)
for k,c,v,i in rows: d[i][c]=d[i].get(c,0)+v
ids=sorted(d)
for i in ids[-10:]:
    x=d[i]; w=x.get('SQ_WAVES',1)
    print("  valu/wave %.0f salu/wave %.0f smem/wave %.0f waves %d"%(x.get('SQ_INSTS_VALU',0)/w, x.get('SQ_INSTS_SALU',0)/w, x.get('SQ_INSTS_SMEM',0)/w, w))
PY
done
rm -rf /tmp/k2; timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d /tmp/k2 -o k -- python bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-all-cores --no-cpu-baseline > /tmp/k2.log 2>&1
python - <<PY
import sqlite3,glob,collections
db=glob.glob("/tmp/k2/*.db")[0]
cur=sqlite3.connect(db).cursor()
rows=list(cur.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection where kernel_name like '%k_icp_step<%' order by dispatch_id"))
d=collections.defaultdict(dict)
for k,c,v,i in rows: d[i][c]=d[i].get(c,0)+v
for i in sorted(d)[-3:]: print(d[i])
PY
