#!/bin/bash
# round-2 GPU session K: SN backward tiles, pipelined BN / colsum loads, vector max-pool: kernel + network parity, step bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2k
mkdir -p $O
( time timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py tests/test_fullwidth_gpu.py -k "batchnorm or pool or attention_core or discriminator_fwd_bwd or generator_fwd_bwd or (biggan128w and golden)" -m gpu -q --maxfail=40 -p no:cacheprovider --durations=5 ) > $O/pytest_gpu.txt 2>&1
tail -12 $O/pytest_gpu.txt
timeout 500 python bench.py --steps 8 --warmup 3 --fid-samples 0 --no-cpu-baseline --no-extras > $O/bench_step.json 2> $O/bench_step.err
tail -c 2200 $O/bench_step.json
