#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3final; mkdir -p $O
python bench.py > $O/bench_default.json 2>$O/bench_default.err
python bench.py --workload c2 > $O/bench_c2.json 2>$O/bench_c2.err
python bench.py --workload c3 --no-cpu-baseline > $O/bench_c3.json 2>$O/bench_c3.err
python bench.py --workload c4 --no-cpu-baseline > $O/bench_c4.json 2>$O/bench_c4.err
python bench.py --workload c4 --batch 256 --no-cpu-baseline > $O/bench_c4_256.json 2>>$O/bench_c4.err
timeout 900 python bench.py --workload c5 > $O/bench_c5.json 2>$O/bench_c5.err
for f in bench_default bench_c2 bench_c3 bench_c4 bench_c4_256; do python - $O/$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
print(sys.argv[1].split('/')[-1], round(d['value']), round(d['ms_per_step'],4), 'frac', round(r.get('frac',0),4), 'traffic', r.get('traffic'), (d.get('cpu_baseline') or {}).get('value'))
for k in ('c3','c4_256','c4_32','c4_8','c5'):
    if k in d: print('   ', k, round(d[k]['value'],1), round(d[k]['ms_per_step'],4), round(d[k]['roofline']['frac'],4), d[k]['roofline'].get('traffic'))
PY
done
cut -c1-300 $O/bench_c5.json
