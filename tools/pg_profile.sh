#!/bin/bash
# per-kernel, per-grid durations of one C5 solve (HIP graph off: every kernel a dispatch of its own)
#   usage: gpurun -- 'bash tools/pg_profile.sh <tag> [ENV=VALUE ...]'
TAG=${1:-pg}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for kv in "$@"; do export "$kv"; done
rm -rf /tmp/tr_c5
SRRG2_AMD_PG_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_c5 -o t -- python $R/tools/bench_posegraph.py > $O/bench_posegraph_traced.json 2> $O/rocprof.err
DB=$(find /tmp/tr_c5 -name '*.db' | head -1)
python $R/tools/pg_trace.py $DB > $O/trace_c5_per_grid.txt 2>&1
python $R/tools/rocpd_summary.py $O/rocprofv3_c5_summary.txt kernel_trace_stats=$DB > /dev/null 2>&1
grep -v rocprim $O/trace_c5_per_grid.txt | head -${LINES_OUT:-70}
