#!/bin/bash
# round-3 GPU session K: the multi-rank bench path on one device (2 gloo ranks on cuda:0): self-launch, early exchange, sync-BN, FID leg with 2 ranks
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r3k
mkdir -p $O
( time SG_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 2 --batch 128 --no-cpu-baseline --no-extras --fid-samples 2048 ) > $O/bench_2ranks_one_device.json 2> $O/bench_2ranks_one_device.err
tail -c 1500 $O/bench_2ranks_one_device.json; echo; grep -v "^$" $O/bench_2ranks_one_device.err | tail -12 | cut -c1-300
( time timeout 100 python bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_2ranks_refused.json 2> $O/bench_2ranks_refused.err; echo "rc=$?"
tail -5 $O/bench_2ranks_refused.err | cut -c1-300
