cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2zp; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python bench.py --workload c2 --no-cpu-baseline 2>/dev/null | cut -c1-170
python bench.py --workload c4 --no-cpu-baseline 2>/dev/null | cut -c1-170
