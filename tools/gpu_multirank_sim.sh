#!/bin/bash
# exercises bench.py's N>1 control flow on a 1-GPU box (ranks share GPU 0, gloo instead of RCCL)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
CURVIS_BENCH_SHARE_DEVICE=1 CURVIS_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 4 --warmup 1 --sky 2048 > $OUT/multirank_sim.log 2>&1
echo "rc=$?" >> $OUT/multirank_sim.log
# and the real backend with a single rank under the launcher (WORLD_SIZE=1)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29578 bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/singlerank_launcher.log 2>&1
echo "rc=$?" >> $OUT/singlerank_launcher.log
