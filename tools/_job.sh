cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3l; mkdir -p $O
python bench.py --no-cpu-baseline > $O/b3.json 2>/dev/null
python - $O/b3.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("c2", d["ms_per_step"], d["step_ms_min_median_max"])
for k in ("c3","c4_256","c4_32","c4_8"):
    r=d.get(k); print(k, round(r["value"],1), round(r["ms_per_step"],4), r["step_ms_min_median_max"])
PY
