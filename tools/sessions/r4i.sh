#!/bin/bash
# round-4 GPU session I: teacher-forced block-level bf16 comparisons (width-8 and full-width fixtures)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4i
mkdir -p $O
rm -f gpurun_out/fullwidth_parity.txt
( time timeout 800 python -m pytest tests/test_blocks_gpu.py tests/test_fullwidth_gpu.py -q -p no:cacheprovider --maxfail=30 --durations=8 -k "bf16_vs_emulating and (biggan or bigdeep)" 2>&1 | grep -E "teacher|FAIL|passed|failed|Error|error|s call" | cut -c1-230 | tail -150 ) > $O/pytest_teacher.txt 2>&1
cat $O/pytest_teacher.txt
