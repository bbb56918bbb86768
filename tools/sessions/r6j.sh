#!/bin/bash
# round 6: the sharded optimizer step (reduce-scatter -> Adam on 1/world -> all-gather) against the all-reduce step, and the whole two-rank file
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6j; mkdir -p $O
( time timeout 900 python -m pytest tests/test_dist_gpu.py -q -m gpu -p no:cacheprovider -s 2>&1 | grep -vE "^(D grad|G grad|state|early)" | tail -40 ) > $O/pytest_dist.txt 2>&1
grep -E " passed| failed|^FAILED|^ERROR|Error|not bit-identical" $O/pytest_dist.txt | head -20
