#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r3q
O=$PWD/gpurun_out/r3q
timeout 900 python -m pytest tests/test_gpu_posegraph.py tests/test_gpu_graph_lifecycle.py -m gpu -x -q > $O/pytest_pg.log 2>&1
tail -3 $O/pytest_pg.log
SRRG2_AMD_PG_DEBUG=1 timeout 300 python tools/bench_posegraph.py > $O/bench_pg_debug.log 2>&1
grep "incidences" $O/bench_pg_debug.log
grep "built in" $O/bench_pg_debug.log | cut -c280-400
for i in 1 2; do timeout 300 python tools/bench_posegraph.py 2>&1 | cut -c1-200; done
