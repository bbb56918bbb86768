#!/bin/bash
# Round 5: first GPU run of the ADA pipeline (csrc/ext/ada.hip behind studiogan_amd.ada_aug) with the rest of tests/test_aug_gpu.py, and its timing
mkdir -p gpurun_out/r5o
( time timeout 200 python -m pytest tests/test_aug_gpu.py -x -q ) > gpurun_out/r5o/pytest_aug_gpu.txt 2>&1; echo "rc=$?" >> gpurun_out/r5o/pytest_aug_gpu.txt
tail -6 gpurun_out/r5o/pytest_aug_gpu.txt
timeout 100 python tools/aug_bench.py > gpurun_out/r5o/aug_bench.txt 2>&1; echo "rc=$?" >> gpurun_out/r5o/aug_bench.txt
grep -E "^---|AdaAugment|rc=" gpurun_out/r5o/aug_bench.txt
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r5o/prof -o ada -- python $GRAFT_REPO_ROOT/tools/aug_bench.py > /dev/null 2>&1
