#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r3g
mkdir -p $O
( time timeout 300 python -m pytest tests/test_dist_gpu.py -m gpu -q -p no:cacheprovider -k "final_when_sent" -s ) > $O/pytest_selftest.txt 2>&1
grep -E "RuntimeError|received gradient|FAILED|passed|failed" $O/pytest_selftest.txt | cut -c1-400 | head -20
( time timeout 200 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "spectral or sn_ or (biggan32 and golden) or (sngan32 and golden)" ) > $O/pytest_sn.txt 2>&1
grep -E "FAILED|passed|failed" $O/pytest_sn.txt | cut -c1-200
