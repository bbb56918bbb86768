#!/bin/bash
# fifth session: batch-norm apply policy (non-temporal + 4096 workgroups on tensors beyond the Infinity Cache, streaming kernels down to 4 x 4 maps): tests, tools/bn_bench.py, same-box step time
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r7g; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_blocks_gpu.py -q -m gpu -rf -p no:cacheprovider 2>&1 | grep -E "^FAILED|passed|failed" | tail -8
timeout 300 python tools/bn_bench.py --variants ,02 2>&1 | grep -v amdgpu.ids | tee $O/bn_bench.txt
B="python bench.py --steps 10 --warmup 3 --no-extras --fid-samples 0 --no-cpu-baseline"
for z in 1 2; do
  timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('auto', d['ms_per_step'], d['roofline_hbm']['batch_norm'])"
  SG_BN_APPLY=02 timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('old ', d['ms_per_step'], d['roofline_hbm']['batch_norm'])"
done
