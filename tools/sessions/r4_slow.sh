#!/bin/bash
# the slow full-width comparisons `pytest -m gpu` skips (tests/test_fullwidth_gpu.py `slow`): batch curve, 256 x 256 BigGAN-deep, stage-wise WGAN-GP / BigGAN-deep
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_slow
mkdir -p $O
rm -f gpurun_out/fullwidth_parity.txt
( time SG_SLOW=1 timeout 1100 python -m pytest tests/test_fullwidth_gpu.py -q -p no:cacheprovider --maxfail=20 --durations=10 -k "batch_curve or bigdeep256w or (stagewise and (wgangp128w or bigdeep128w))" 2>&1 | tail -30 ) > $O/pytest_slow.txt 2>&1
cat $O/pytest_slow.txt | cut -c1-220
cp gpurun_out/fullwidth_parity.txt $O/ 2>/dev/null
