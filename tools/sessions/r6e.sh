#!/bin/bash
# round 6: per-tensor bf16 step tables against the reference's golden vectors (to set measured first-forward bounds), then the WHOLE GPU suite with durations
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6e; mkdir -p $O
( time timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fullwidth_gpu.py -q -m gpu -p no:cacheprovider -s -k "step_vs_golden and True" 2>&1 ) > $O/pytest_bf16_steps.txt 2>&1
grep -E " passed| failed|^FAILED|^ERROR" $O/pytest_bf16_steps.txt | head
( time timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=40 2>&1 | tail -80 ) > $O/pytest_gpu_full.txt 2>&1
grep -E " passed| failed|^FAILED|^ERROR|real" $O/pytest_gpu_full.txt | head
