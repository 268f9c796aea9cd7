#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_posegraph.py tests/test_gpu_graph_lifecycle.py tests/test_gpu_full_size.py -k "posegraph or graph or c5 or lifecycle or pose" -m gpu -x -q 2>&1 | tail -2
for e in 0 1 0 1; do SRRG2_AMD_PG_NODE_ROWS=$e timeout 300 python tools/bench_posegraph.py 2>&1 | cut -c1-230; done
