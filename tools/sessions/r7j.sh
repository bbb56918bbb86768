#!/bin/bash
# fifth session: attention forward software-pipelined over the 32-key blocks (PV of block i-1 and scores of block i+1 issued in front of the softmax arithmetic of block i); kernel tests, same-box A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r7j; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -rf -p no:cacheprovider -k "attention or attn" 2>&1 | grep -E "^FAILED|passed|failed" | tail -5
for z in 1 2; do
  echo "== new"; timeout 200 python tools/attn_bench.py --iters 20 2>&1 | grep -v amdgpu.ids | tee -a $O/attn_new.txt
  echo "== old"; SG_LIBSGAMD=pytorch-studiogan_amd/libsgamd_attnold.so timeout 200 python tools/attn_bench.py --iters 20 2>&1 | grep -v amdgpu.ids | tee -a $O/attn_old.txt
done
