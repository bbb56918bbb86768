#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6w; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_blocks_gpu.py tests/test_heads_gpu.py tests/test_dist_gpu.py -x -q -m gpu 2>&1 | tail -3
for z in 0 1 0 1; do
  SG_ZERO_POOL=$z timeout 200 python tools/extra_run.py wgangp128_bs64_bf16 3 2>/dev/null | grep -o '"images_per_sec": [0-9.]*' | sed "s/^/wgangp pool=$z /"
done
B="python bench.py --steps 10 --warmup 3 --no-extras --fid-samples 0 --no-cpu-baseline"
for z in 0 1 0 1; do
  SG_ZERO_POOL=$z timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3 pool=$z', d['ms_per_step'])"
done
