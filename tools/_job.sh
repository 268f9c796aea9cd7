cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2zu; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python tools/bench_scene.py 2>/dev/null | cut -c1-900
python - <<'PY'
# many-to-one merge: 200k correspondences onto 20k scene points (10 per scene point), timing of the merge
import time, numpy as np, sys
sys.path.insert(0, '.')
import srrg2_slam_interfaces_amd as pkg
from srrg2_slam_interfaces_amd import mapping
import inspect
print([n for n in dir(mapping) if not n.startswith('_')][:20])
PY
