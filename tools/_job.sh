cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3j; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_parity.txt 2>&1; tail -3 $O/pytest_parity.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
A=$O/ab_fused.txt; rm -f $A
bash tools/ab_env.sh $A "--workload c2" "SRRG2_AMD_FUSED_CONTROL=0" "SRRG2_AMD_FUSED_CONTROL=1" "SRRG2_AMD_FUSED_CONTROL=0" "SRRG2_AMD_FUSED_CONTROL=1"
bash tools/ab_env.sh $A "--workload c3" "SRRG2_AMD_FUSED_CONTROL=0" "SRRG2_AMD_FUSED_CONTROL=1"
bash tools/ab_env.sh $A "--workload c4 --batch 32" "SRRG2_AMD_FUSED_CONTROL=0" "SRRG2_AMD_FUSED_CONTROL=1"
bash tools/ab_env.sh $A "--workload c4 --batch 8" "SRRG2_AMD_FUSED_CONTROL=0" "SRRG2_AMD_FUSED_CONTROL=1"
bash tools/ab_env.sh $A "--workload c4 --batch 256 --steps 10" "SRRG2_AMD_FUSED_CONTROL=0" "SRRG2_AMD_FUSED_CONTROL=1"
bash tools/ab_env.sh $A "--workload c2 --points 10000" "SRRG2_AMD_FUSED_CONTROL=0" "SRRG2_AMD_FUSED_CONTROL=1"
cat $A
