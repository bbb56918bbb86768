#!/bin/bash
# Round 5, last GPU call (8.6 GPU-minutes left): first run of the csrc/ext kernels on the GPU -- their parity tests (already green on the CPU interpreter),
# then their microbenchmark. Nothing of the benchmarked step is rebuilt or re-measured here.
mkdir -p gpurun_out/r5j
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
( time timeout 280 python -m pytest tests/test_aug_gpu.py -x -q ) > gpurun_out/r5j/pytest_aug_gpu.txt 2>&1
echo "rc=$?" >> gpurun_out/r5j/pytest_aug_gpu.txt
tail -5 gpurun_out/r5j/pytest_aug_gpu.txt
timeout 90 python tools/aug_bench.py > gpurun_out/r5j/aug_bench.txt 2>&1
echo "rc=$?" >> gpurun_out/r5j/aug_bench.txt
cat gpurun_out/r5j/aug_bench.txt
( time timeout 120 python -m pytest tests/test_model_gpu.py -x -q -k "biggan32 or sngan32" ) > gpurun_out/r5j/pytest_model_subset.txt 2>&1
echo "rc=$?" >> gpurun_out/r5j/pytest_model_subset.txt
tail -3 gpurun_out/r5j/pytest_model_subset.txt
