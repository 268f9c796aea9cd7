cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3i; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
SRRG2_AMD_LDS_TILE=0 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
A=$O/ab.txt; rm -f $A
bash tools/ab_env.sh $A "--workload c4 --batch 256 --steps 10" "-" "SRRG2_AMD_LDS_TILE=0"
bash tools/ab_env.sh $A "--workload c4 --batch 32" "-" "SRRG2_AMD_LDS_TILE=0"
bash tools/ab_env.sh $A "--workload c4 --batch 8" "-" "SRRG2_AMD_LDS_TILE=0"
bash tools/ab_env.sh $A "--workload c2" "-" "SRRG2_AMD_QUEUE_MIN=1000000"
bash tools/ab_env.sh $A "--workload c2 --points 30000" "-" "SRRG2_AMD_LDS_TILE=0"
bash tools/ab_env.sh $A "--workload c3" "-"
cat $A
