#!/bin/bash
# Same-box A/B of strategy knobs: runs `bench.py <args>` once per environment setting and prints value / ms_per_step.
#   usage: bash tools/ab_env.sh <out-file> "<bench args>" "VAR=a" "VAR=b VAR2=c" ...   ("-" = no variables)
OUT=$1; ARGS=$2; shift 2
cd ${GRAFT_REPO_ROOT:-.}
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
  line=$(env $envs python bench.py $ARGS --no-cpu-baseline 2>/dev/null | tail -1)
  python - "$cfg" "$ARGS" "$line" >> $OUT <<'PY'
import json, sys
cfg, args, line = sys.argv[1:4]
try:
    d = json.loads(line)
    print("%-44s | %-28s | %10.0f it/s  %8.4f ms/step  kernel %.4f ms  all_success=%s" % (cfg, args, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["config"]["all_success"]))
except Exception as e:
    print("%-44s | %-28s | FAILED %r %s" % (cfg, args, e, line[:200]))
PY
done
