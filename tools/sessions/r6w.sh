#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6w; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_blocks_gpu.py tests/test_dist_gpu.py -q -m gpu -rf 2>&1 | grep -E "^FAILED|passed|failed" | tail -5
B="python bench.py --steps 10 --warmup 3 --no-extras --fid-samples 0 --no-cpu-baseline"
for z in 1 2 3; do
  timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3', d['ms_per_step'])"
done
