#!/bin/bash
# fifth session: batch-norm apply pass variants (loads in flight, non-temporal accesses, grid size): tools/bn_bench.py
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r7f; mkdir -p $O
timeout 600 python tools/bn_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/bn_bench.txt
