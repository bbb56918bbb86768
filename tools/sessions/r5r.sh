#!/bin/bash
# Round 5, the last GPU seconds: the InfoGAN cases (they also run the three generator backbones and every discriminator head path changed after r5p)
mkdir -p gpurun_out/r5r
( time timeout 28 python -m pytest tests/test_wide_info_gpu.py tests/test_wide_ada_gpu.py -x -q ) > gpurun_out/r5r/pytest_wide.txt 2>&1; echo "rc=$?" >> gpurun_out/r5r/pytest_wide.txt
tail -5 gpurun_out/r5r/pytest_wide.txt
