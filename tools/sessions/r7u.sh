#!/bin/bash
# fifth session: bench.py two-rank path end to end on one device after the changes to its profiling legs and the pinned table uploads (plumbing, not a scaling number); two-rank GPU tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r7u; mkdir -p $O
for sc in "" "--comm torch"; do
  tag=p2p; [ -n "$sc" ] && tag=torch
  ( SG_BENCH_ONE_DEVICE=1 timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 3 --warmup 1 --batch 128 --no-cpu-baseline --no-extras --fid-samples 0 $sc ) > $O/bench_2ranks_one_device_$tag.json 2> $O/bench_2ranks_one_device_$tag.err
  echo "2 ranks one device $tag: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_2ranks_one_device_$tag.json | head -1) $(grep -o '"exposed_comm_ms_per_step": [0-9.]*' $O/bench_2ranks_one_device_$tag.json | head -1) $(grep -o '"sync_bn_exchange": "[^"]*"' $O/bench_2ranks_one_device_$tag.json)"; tail -2 $O/bench_2ranks_one_device_$tag.err | cut -c1-300
done
timeout 600 python -m pytest tests/test_dist_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tail -2
