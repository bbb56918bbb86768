#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r3f
mkdir -p $O
( time timeout 300 python -m pytest tests/test_dist_gpu.py -m gpu -q -p no:cacheprovider -k "early" -s ) > $O/pytest_early.txt 2>&1
grep -E "worst relative|FAILED|passed|failed" $O/pytest_early.txt | cut -c1-220
( time timeout 200 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "spectral or sn_ or (biggan32 and golden)" ) > $O/pytest_sn.txt 2>&1
grep -E "FAILED|passed|failed" $O/pytest_sn.txt | cut -c1-200
