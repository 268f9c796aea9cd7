#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r3q
O=$PWD/gpurun_out/r3q
rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_posegraph.py tests/test_gpu_graph_lifecycle.py -m gpu -x -q > $O/pytest_pg.log 2>&1
tail -3 $O/pytest_pg.log
for i in 1 2; do timeout 300 python tools/bench_posegraph.py >> $O/bench_pg.log 2>&1; done
cat $O/bench_pg.log | cut -c1-330
export TMPDIR=/tmp
R=$PWD
cd /tmp
SRRG2_AMD_PG_GRAPH=0 timeout 600 rocprofv3 --kernel-trace -d $O/pgtrace -o pg -- python $R/tools/bench_posegraph.py > /dev/null 2>&1
DB=$(find $O/pgtrace -name "*.db" | head -1)
python $R/tools/pg_trace.py $DB > $O/pg_trace.txt
head -24 $O/pg_trace.txt | cut -c1-120
rm -rf $O/pgtrace
