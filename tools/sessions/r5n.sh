#!/bin/bash
# Round 5: GPU run of tests/test_aug_gpu.py after APA / weight clipping joined it (csrc/ext/regularisers.hip)
mkdir -p gpurun_out/r5n
( time timeout 200 python -m pytest tests/test_aug_gpu.py -x -q ) > gpurun_out/r5n/pytest_aug_gpu.txt 2>&1; echo "rc=$?" >> gpurun_out/r5n/pytest_aug_gpu.txt
tail -6 gpurun_out/r5n/pytest_aug_gpu.txt
