#!/bin/bash
# round-2 GPU session C: new components (conditioning heads, device PRDC / sqrtm, split-K reduce rewrite, dist test fix) + the full default bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2c
mkdir -p $O
( time timeout 900 python -m pytest tests/test_heads_gpu.py tests/test_eval_gpu.py tests/test_dist_gpu.py tests/test_conv_v2_gpu.py tests/test_kernels_gpu.py \
    "tests/test_model_gpu.py::test_training_step_vs_golden" tests/test_blocks_gpu.py::test_gradient_penalty_double_backward \
    -m gpu -q --maxfail=80 -p no:cacheprovider --durations=10 ) > $O/pytest_gpu.txt 2>&1
tail -30 $O/pytest_gpu.txt
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -5 $O/bench_default.err
tail -c 6000 $O/bench_default.json
