#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$R/gpurun_out/r4l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_posegraph.py tests/test_gpu_graph_lifecycle.py tests/test_cpp_mirror.py tests/test_gpu_full_size.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
python bench.py > $O/bench_default.json 2>$O/bench_default.err; python - $O/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('C2', round(d['value']), d['ms_per_step'], 'long', d.get('value_long'), d['roofline']['frac'], d.get('cpu_baseline',{}).get('value'))
for k in ('c3','c4_256','c4_32','c4_8'):
    print(k, round(d[k]['value']), d[k]['ms_per_step'], d[k]['roofline']['frac'])
print('c5', d['c5']['ms_per_step'], d['c5'].get('warm'), d['c5'].get('cpu_baseline'))
print('proj', d.get('projected_8gpu_speedup'), d.get('nested_error'))
PY
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d /tmp/pmc_c3_$c -o p -- python $R/bench.py --workload c3 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc $c -d /tmp/pmc_c2_$c -o p -- python $R/bench.py --workload c2 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc $c -d /tmp/pmc_c4_$c -o p -- python $R/bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc $c -d /tmp/pmc_c4_256_$c -o p -- python $R/bench.py --workload c4 --batch 256 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
python $R/tools/traffic_from_pmc.py $O/traffic_c3.json c3 $(find /tmp/pmc_c3_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/pmc_c3_WRITE_SIZE -name '*.db' | head -1) > /dev/null
python $R/tools/traffic_from_pmc.py $O/traffic_c2.json c2 $(find /tmp/pmc_c2_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/pmc_c2_WRITE_SIZE -name '*.db' | head -1) > /dev/null
python $R/tools/traffic_from_pmc.py $O/traffic_c4.json c4 $(find /tmp/pmc_c4_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/pmc_c4_WRITE_SIZE -name '*.db' | head -1) 32 > /dev/null
python $R/tools/traffic_from_pmc.py $O/traffic_c4_256.json c4 $(find /tmp/pmc_c4_256_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/pmc_c4_256_WRITE_SIZE -name '*.db' | head -1) 256 > /dev/null
for f in c2 c3 c4 c4_256; do python -c "import json;d=json.load(open('$O/traffic_$f.json'));print('$f',d['bytes_per_slice_pass'],{k:(round(v['fetch_bytes_corrected']/1e6,1),round(v['write_bytes']/1e6,1),v['dispatches']) for k,v in d['kernels'].items()})"; done
