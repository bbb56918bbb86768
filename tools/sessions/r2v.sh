#!/bin/bash
# round-2 GPU session V: the ResNet / SNGAN configurations (64-cout layers now on conv_v2's partial-tile path) through the network parity tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2v
mkdir -p $O
( time timeout 600 python -m pytest tests/test_fullwidth_gpu.py tests/test_blocks_gpu.py tests/test_model_gpu.py -k "(wgangp128w and golden) or sngan32w or resgan32 or sngan32 or wgangp32" -m gpu -q --maxfail=40 -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
tail -6 $O/pytest_gpu.txt
