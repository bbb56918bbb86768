#!/bin/bash
# round-2 GPU session G: straight-line taps in the halo forward kernel (compile-time sub-step count, branch-free DMA)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2g
mkdir -p $O
( time timeout 900 python -m pytest tests/test_conv_v2_gpu.py tests/test_kernels_gpu.py -k "conv or attention or attn" -m gpu -q --maxfail=40 -p no:cacheprovider --durations=5 ) > $O/pytest_gpu.txt 2>&1
tail -12 $O/pytest_gpu.txt
timeout 300 python tools/conv_bench.py > $O/conv_layer_table.txt 2> $O/conv_layer_table.err
cat $O/conv_layer_table.txt
timeout 500 python bench.py --steps 8 --warmup 3 --fid-samples 0 --no-cpu-baseline --no-extras > $O/bench_step.json 2> $O/bench_step.err
tail -c 2200 $O/bench_step.json
