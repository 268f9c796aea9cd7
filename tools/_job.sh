set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2n; mkdir -p $O
(time timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -8) > $O/pytest.log 2>&1
timeout 600 python tools/bench_small.py > $O/bench_small.json 2> $O/bench_small.err
SRRG2_AMD_FAST_MIN=0 timeout 600 python tools/bench_small.py > $O/bench_small_fastmin0.json 2>> $O/bench_small.err
timeout 600 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
timeout 600 python bench.py --workload c4 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
timeout 600 python bench.py --workload c4 --batch 256 --steps 5 --warmup 2 > $O/bench_c4_256.json 2> $O/bench_c4_256.err
