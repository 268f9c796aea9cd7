#!/bin/bash
# scratch job file for `gpurun -- 'bash tools/_job.sh'` (rewritten per experiment; see tools/profile_round.sh for the
# end-of-round measurement)
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -c "import __graft_entry__ as g; g.smoke()"
