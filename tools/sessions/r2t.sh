#!/bin/bash
# round-2 GPU session T: plumbing check of bench.py's multi-rank path on one GPU (2 ranks on cuda:0, gloo) -- never a benchmark configuration
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2t
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time SG_BENCH_ONE_DEVICE=1 timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --fid-samples 1024 --no-cpu-baseline ) > $O/bench_2ranks_one_device.json 2> $O/bench_2ranks_one_device.err
tail -5 $O/bench_2ranks_one_device.err
tail -c 1200 $O/bench_2ranks_one_device.json
