#!/bin/bash
# scratch: the command file of the last `gpurun -- 'bash tools/_job.sh'` call of the session (GPU tests + default bench)
cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 300 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py 2>/dev/null | tail -1 | cut -c1-400
