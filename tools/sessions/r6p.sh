#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6p; mkdir -p $O
timeout 300 python tools/quad_bench.py > $O/quad_bench.txt 2>&1; cut -c1-220 $O/quad_bench.txt
