#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r3q
O=$PWD/gpurun_out/r3q
SRRG2_AMD_PG_DEBUG=1 timeout 300 python tools/bench_posegraph.py > $O/bench_pg_debug.log 2>&1
grep "captured" $O/bench_pg_debug.log
