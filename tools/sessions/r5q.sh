#!/bin/bash
# Round 5, last GPU seconds: tests/test_aug_gpu.py on the final host code (the ADA mirror composes its 3 x 3 / 4 x 4 matrices without library GEMMs since r5o)
mkdir -p gpurun_out/r5q
( time timeout 50 python -m pytest tests/test_aug_gpu.py -x -q ) > gpurun_out/r5q/pytest_aug_gpu.txt 2>&1; echo "rc=$?" >> gpurun_out/r5q/pytest_aug_gpu.txt
tail -5 gpurun_out/r5q/pytest_aug_gpu.txt
