#!/bin/bash
# round-4 GPU session B: first execution of the quad kernels (conv_q.h / wgrad_q.h / conv_q.hip) against torch on the CPU
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4b
mkdir -p $O
( time timeout 600 python -m pytest tests/test_quad_gpu.py -q -p no:cacheprovider --maxfail=30 2>&1 | tail -60 ) > $O/pytest_quad.txt 2>&1
cat $O/pytest_quad.txt | cut -c1-250
