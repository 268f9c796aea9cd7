cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2zx; mkdir -p $O
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 tools/bench_point_shard.py --points 2000000 --steps 10 2>$O/e1.txt | tee $O/point_shard_1rank_rccl.json | cut -c1-600; tail -3 $O/e1.txt
SRRG2_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 tools/bench_point_shard.py --points 2000000 --steps 10 2>$O/e2.txt | tee $O/point_shard_2ranks_shared.json | cut -c1-600; tail -3 $O/e2.txt
