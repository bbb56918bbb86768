#!/bin/bash
# round 6: sync-BN's exchange fused into the finalize kernel over IPC-mapped peer mailboxes -- two processes on one GPU
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6h; mkdir -p $O
( time timeout 600 python -m pytest tests/test_dist_gpu.py -q -m gpu -p no:cacheprovider -x -k "peer_store" 2>&1 | tail -40 ) > $O/pytest_p2p.txt 2>&1
grep -E " passed| failed|^FAILED|^ERROR|Error|error" $O/pytest_p2p.txt | head -20
