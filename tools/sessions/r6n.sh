#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6n; mkdir -p $O
( time timeout 900 python -m pytest tests/test_blocks_gpu.py tests/test_style_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -8 ) > $O/pytest.txt 2>&1
grep -E " passed| failed|^FAILED|^ERROR" $O/pytest.txt | head
