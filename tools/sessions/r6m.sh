#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6m; mkdir -p $O
timeout 300 python tools/conv_bench.py > $O/conv_layer_table.txt 2>&1; cat $O/conv_layer_table.txt | cut -c1-200
