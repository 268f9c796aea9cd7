cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2ze; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
SRRG2_AMD_PG_DEBUG=1 timeout 600 python tools/bench_posegraph.py --cpu > $O/bench_c5.json 2>$O/bench_c5.err; cut -c1-330 $O/bench_c5.json; tail -1 $O/bench_c5.err | cut -c250-
SRRG2_AMD_PG_PASSES=4 timeout 600 python tools/bench_posegraph.py > $O/bench_c5_passes4.json 2>/dev/null; cut -c100-230 $O/bench_c5_passes4.json
python bench.py --workload c2 --no-cpu-baseline > $O/bench_c2.json 2>$O/bench_c2.err; cut -c1-200 $O/bench_c2.json
