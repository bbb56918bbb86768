#!/bin/bash
# fifth session: the full-width comparisons `pytest -m gpu` skips (SG_SLOW=1) on the final code
# biggan128w / wgangp128w / bigdeep128w / bigdeep256w, the fp32 step of bigdeep128w, the bf16 batch curves
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r7_slow
mkdir -p $O
rm -f gpurun_out/fullwidth_parity.txt
( time SG_SLOW=1 timeout 840 python -m pytest tests/test_fullwidth_gpu.py -q -p no:cacheprovider --maxfail=20 --durations=12 -k "batch_curve or stagewise or (step_vs_golden and bigdeep128w)" 2>&1 | tail -32 ) > $O/pytest_slow.txt 2>&1
cat $O/pytest_slow.txt | cut -c1-220
cp gpurun_out/fullwidth_parity.txt $O/ 2>/dev/null
