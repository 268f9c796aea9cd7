#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3x; mkdir -p $O; rm -f $O/ab.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tuning.py tests/test_gpu_full_size.py tests/test_loop_detector.py -m gpu -x -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for w in "--workload c4 --batch 32 --steps 10" "--workload c4 --batch 256 --steps 5" "--workload c4 --batch 8 --steps 10"; do
  bash tools/ab_env.sh $O/ab.txt "$w" "SRRG2_AMD_TUNE=8388608" "-" "SRRG2_AMD_TUNE=8388608" "-"
done
cat $O/ab.txt
export TMPDIR=/tmp; R=$PWD; cd /tmp
for t in 0 8388608; do
rm -rf /tmp/trk; SRRG2_AMD_TUNE=$t timeout 300 rocprofv3 --kernel-trace -d /tmp/trk -o t -- python $R/bench.py --workload c4 --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/iter_durations.py $(find /tmp/trk -name '*.db' | head -1) 10 | head -1
done
