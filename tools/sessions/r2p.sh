#!/bin/bash
# round-2 GPU session P: vector elementwise kernels + conv_v3 small-tile rule: parity on the deep / ResNet configurations, extras re-measured
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2p
mkdir -p $O
( time timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py tests/test_fullwidth_gpu.py tests/test_conv_v2_gpu.py -k "pool_misc or bigdeep or (wgangp128w and golden) or conv_v3" -m gpu -q --maxfail=40 -p no:cacheprovider --durations=5 ) > $O/pytest_gpu.txt 2>&1
tail -12 $O/pytest_gpu.txt
for W in bigdeep128_bs256_bf16 wgangp128_bs64_bf16 sngan32_bs256_fp32; do
  timeout 300 python tools/extra_run.py $W 3 > $O/extra_$W.json 2> $O/extra_$W.err
  tail -1 $O/extra_$W.json | cut -c1-600
done
timeout 500 python bench.py --steps 8 --warmup 3 --fid-samples 0 --no-cpu-baseline --no-extras > $O/bench_step.json 2> $O/bench_step.err
tail -c 700 $O/bench_step.json
