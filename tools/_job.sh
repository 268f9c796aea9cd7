#!/bin/bash
# scratch job for gpurun
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/t14; mkdir -p $O
python -m pytest tests/test_gpu_posegraph.py -x -q 2>&1 | tail -3 > $O/pytest_pg.txt
for lag in 0 0.05 0.2; do
SRRG2_AMD_PG_LAG=$lag python bench.py --workload c5 > $O/bench_c5_lag$lag.json 2>$O/bench_c5.err
done
