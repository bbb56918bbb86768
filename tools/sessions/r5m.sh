#!/bin/bash
# Round 5: GPU run of the second-order cases added after r5l (R1 / maxGP on BigGAN-deep, SN-DCGAN, ResNet-GAN discriminators; WGAN-GP penalty through attention in fp32 and bf16)
mkdir -p gpurun_out/r5m
( time timeout 300 python -m pytest tests/test_blocks_gpu.py -x -q -k "r1_and_maxgp or gradient_penalty" ) > gpurun_out/r5m/pytest_second_order.txt 2>&1; echo "rc=$?" >> gpurun_out/r5m/pytest_second_order.txt
tail -6 gpurun_out/r5m/pytest_second_order.txt
