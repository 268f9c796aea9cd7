set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2h; mkdir -p $O
(time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15) > $O/pytest.log 2>&1
B='import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"])'
for g in 1 0; do for ppt in 1 2; do
echo "== GATHER=$g PPT=$ppt c4" >> $O/sweep.log
SRRG2_AMD_FAST_GATHER=$g SRRG2_AMD_FAST_PPT=$ppt timeout 300 python bench.py --workload c4 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "$B" >> $O/sweep.log 2>&1
echo "== GATHER=$g PPT=$ppt c4 256" >> $O/sweep.log
SRRG2_AMD_FAST_GATHER=$g SRRG2_AMD_FAST_PPT=$ppt timeout 300 python bench.py --workload c4 --batch 256 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "$B" >> $O/sweep.log 2>&1
done; done
for g in 0 1; do
echo "== GATHER=$g c2" >> $O/sweep.log
SRRG2_AMD_FAST_GATHER=$g timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "$B" >> $O/sweep.log 2>&1
done
cd /tmp
for ppt in 1 2; do
SRRG2_AMD_FAST_PPT=$ppt timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_c4_$ppt -o t -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/$O/trace_c4_ppt${ppt}_summary.txt trace=$(find /tmp/tr_c4_$ppt -name "*.db" | head -1)
done
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d /tmp/pmc_x -o p -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/$O/pmc.txt pmc=$(find /tmp/pmc_x -name "*.db" | head -1)
