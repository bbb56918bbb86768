#!/bin/bash
# Round 5: GPU check of the attention restructure (pooling as its own autograd nodes, second order through SelfAttention): the first-order attention tests, the
# networks with attention, R1 / maxGP on the BigGAN discriminator, and the C3 step alone for its time.
mkdir -p gpurun_out/r5l
( time timeout 200 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention or attn" ) > gpurun_out/r5l/pytest_attention.txt 2>&1; echo "rc=$?" >> gpurun_out/r5l/pytest_attention.txt
tail -4 gpurun_out/r5l/pytest_attention.txt
( time timeout 200 python -m pytest tests/test_blocks_gpu.py -x -q -k "r1_and_maxgp or biggan32 or bigdeep32" ) > gpurun_out/r5l/pytest_blocks.txt 2>&1; echo "rc=$?" >> gpurun_out/r5l/pytest_blocks.txt
tail -4 gpurun_out/r5l/pytest_blocks.txt
( time timeout 200 python -m pytest tests/test_model_gpu.py -x -q -k "biggan32 or bigdeep32" ) > gpurun_out/r5l/pytest_model.txt 2>&1; echo "rc=$?" >> gpurun_out/r5l/pytest_model.txt
tail -4 gpurun_out/r5l/pytest_model.txt
timeout 150 python bench.py --steps 10 --warmup 3 --no-extras --fid-samples 0 --no-cpu-baseline > gpurun_out/r5l/bench_step_only.json 2> gpurun_out/r5l/bench_step_only.err
python -c "
import json;d=json.loads(open('gpurun_out/r5l/bench_step_only.json').read().strip().splitlines()[-1]);print('ms_per_step',d['ms_per_step'],'value',d['value'])"
