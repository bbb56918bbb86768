#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6b; mkdir -p $O
timeout 300 python tools/diag_logan2.py > $O/diag_logan2.txt 2>&1; tail -12 $O/diag_logan2.txt
