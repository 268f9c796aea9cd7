#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
SRRG2_AMD_HOSTTIME=1 python bench.py --workload c2 --steps 12 --warmup 3 --no-cpu-baseline 2>&1 | grep "compute:" | tail -8
SRRG2_AMD_HOSTTIME=1 python bench.py --workload c3 --steps 6 --warmup 3 --no-cpu-baseline 2>&1 | grep "compute:" | tail -3
SRRG2_AMD_HOSTTIME=1 python bench.py --workload c4 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline 2>&1 | grep "compute:\|upload" | tail -4
