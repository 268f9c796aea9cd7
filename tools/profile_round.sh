#!/bin/bash
# End-of-round measurement on the GPU box: GPU test suite, every bench line, rocprofv3 kernel summaries and the PMC traffic
# passes (separate --pmc runs, never together with a trace).  Everything lands under gpurun_out/$1/; copy what is to be
# judged into profiles/.      usage: gpurun -- 'bash tools/profile_round.sh r2final'
TAG=${1:-round}  # (copy what is to be judged into profiles/r10/)
cd ${GRAFT_REPO_ROOT:-/root/repo}; GRAFT_REPO_ROOT=$(pwd)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python bench.py > $O/bench_default.json 2>$O/bench_default.err
python bench.py --workload c2 > $O/bench_c2.json 2>$O/bench_c2.err
python bench.py --workload c3 --no-cpu-baseline > $O/bench_c3.json 2>$O/bench_c3.err
python bench.py --workload c4 --no-cpu-baseline > $O/bench_c4.json 2>$O/bench_c4.err
python bench.py --workload c4 --batch 256 --no-cpu-baseline > $O/bench_c4_256.json 2>>$O/bench_c4.err
timeout 900 python bench.py --workload c5 > $O/bench_c5.json 2>$O/bench_c5.err
python tools/bench_tracker.py > $O/bench_tracker.json 2>$O/side.err
python tools/bench_small.py > $O/bench_small.json 2>>$O/side.err
python tools/bench_hbst.py > $O/bench_hbst.json 2>>$O/side.err
python tools/bench_scene.py > $O/bench_scene.json 2>>$O/side.err
cd /tmp
for w in c2 c3 c4; do
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_$w -o t -- python $R/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python $R/tools/rocpd_summary.py $O/rocprofv3_${w}_summary.txt kernel_trace_stats=$(find /tmp/tr_$w -name '*.db' | head -1)
done
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_c4_256 -o t -- python $R/bench.py --workload c4 --batch 256 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py $O/rocprofv3_c4_256_summary.txt kernel_trace_stats=$(find /tmp/tr_c4_256 -name '*.db' | head -1)
SRRG2_AMD_PG_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr_c5 -o t -- python $R/tools/bench_posegraph.py > /dev/null 2>&1
python $R/tools/rocpd_summary.py $O/rocprofv3_c5_summary.txt kernel_trace_stats=$(find /tmp/tr_c5 -name '*.db' | head -1)
python $R/tools/pg_trace.py $(find /tmp/tr_c5 -name '*.db' | head -1) > $O/trace_c5_per_grid.txt 2>&1
timeout 600 rocprofv3 --kernel-trace -d /tmp/tr32 -o t -- python $R/bench.py --workload c4 --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/batch_timeline.py $(find /tmp/tr32 -name '*.db' | head -1) > $O/timeline_c4_32_pipelined.txt
python $R/tools/trace_steps.py $(find /tmp/tr_c2 -name '*.db' | head -1) > $O/trace_c2_steps.txt 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d /tmp/pmc_c3_$c -o p -- python $R/bench.py --workload c3 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc $c -d /tmp/pmc_c2_$c -o p -- python $R/bench.py --workload c2 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  SRRG2_AMD_BATCH_PIPELINE=0 timeout 600 rocprofv3 --pmc $c -d /tmp/pmc_c4_$c -o p -- python $R/bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  SRRG2_AMD_BATCH_PIPELINE=0 timeout 600 rocprofv3 --pmc $c -d /tmp/pmc_c4_256_$c -o p -- python $R/bench.py --workload c4 --batch 256 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
python $R/tools/traffic_from_pmc.py $O/traffic_c3.json c3 $(find /tmp/pmc_c3_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/pmc_c3_WRITE_SIZE -name '*.db' | head -1)
python $R/tools/traffic_from_pmc.py $O/traffic_c2.json c2 $(find /tmp/pmc_c2_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/pmc_c2_WRITE_SIZE -name '*.db' | head -1)
python $R/tools/traffic_from_pmc.py $O/traffic_c4.json c4 $(find /tmp/pmc_c4_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/pmc_c4_WRITE_SIZE -name '*.db' | head -1) 32 76800000
python $R/tools/traffic_from_pmc.py $O/traffic_c4_256.json c4 $(find /tmp/pmc_c4_256_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/pmc_c4_256_WRITE_SIZE -name '*.db' | head -1) 256 614400000
# C5: the pose-graph kernels' counters (HIP graph off: every kernel a dispatch of its own)
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVES"; do
  n=$(echo $c | tr ' ' '_')
  SRRG2_AMD_PG_GRAPH=0 timeout 900 rocprofv3 --pmc $c -d /tmp/pmc_c5_$n -o p -- python $R/tools/bench_posegraph.py > /dev/null 2>&1
  python $R/tools/rocpd_summary.py $O/pmc_c5_$n.txt $n=$(find /tmp/pmc_c5_$n -name '*.db' | head -1)
done
PCG=$(python -c "import json; print(sum(json.load(open('$O/bench_c5.json'))['config']['pcg_iterations']))" 2>/dev/null || echo 282)
python $R/tools/traffic_c5_from_pmc.py $O/traffic_c5.json $O/pmc_c5_FETCH_SIZE.txt $O/pmc_c5_WRITE_SIZE.txt $O/pmc_c5_SQ_INSTS_VALU_SQ_WAVES.txt $O/rocprofv3_c5_summary.txt 3 10 $PCG > $O/traffic_c5.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES -d /tmp/pmc_valu -o p -- python $R/bench.py --workload c4 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_summary.py $O/rocprofv3_c4_valu_pmc_summary.txt valu=$(find /tmp/pmc_valu -name '*.db' | head -1)
# the search passes of the 256-alignment batch, pass by pass: durations, instructions, texture-path and LDS activity
timeout 600 rocprofv3 --kernel-trace -d /tmp/tr256 -o t -- python $R/bench.py --workload c4 --batch 256 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
echo "durations" > $O/c4_256_passes.txt; python $R/tools/iter_durations.py $(find /tmp/tr256 -name '*.db' | head -1) 10 >> $O/c4_256_passes.txt
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d /tmp/p256a -o p -- python $R/bench.py --workload c4 --batch 256 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/iter_durations.py $(find /tmp/p256a -name '*.db' | head -1) 10 >> $O/c4_256_passes.txt
timeout 600 rocprofv3 --pmc TA_BUSY_avr SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d /tmp/p256b -o p -- python $R/bench.py --workload c4 --batch 256 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/iter_durations.py $(find /tmp/p256b -name '*.db' | head -1) 10 >> $O/c4_256_passes.txt
for c in FETCH_SIZE WRITE_SIZE; do python $R/tools/iter_durations.py $(find /tmp/pmc_c4_256_$c -name '*.db' | head -1) 10 | grep -v durations >> $O/c4_256_passes.txt; done
cd $R
for f in bench_default bench_c2 bench_c3 bench_c4 bench_c4_256; do python - $O/$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], round(d['value']), d['ms_per_step'], d.get('roofline'), (d.get('cpu_baseline') or {}).get('value'))
PY
done
cut -c1-500 $O/bench_c5.json
