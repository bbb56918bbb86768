#!/bin/bash
# fifth session: attention: lane-half exchange by v_permlane32_swap_b32 instead of ds_bpermute_b32; kernel tests, same-box A/B of tools/attn_bench.py against the previous attn.hip
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r7i; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -rf -p no:cacheprovider -k "attention or attn" 2>&1 | grep -E "^FAILED|passed|failed" | tail -5
for z in 1 2; do
  echo "== new"; timeout 200 python tools/attn_bench.py --iters 20 2>&1 | grep -v amdgpu.ids | tee -a $O/attn_new.txt
  echo "== old"; SG_LIBSGAMD=pytorch-studiogan_amd/libsgamd_attnold.so timeout 200 python tools/attn_bench.py --iters 20 2>&1 | grep -v amdgpu.ids | tee -a $O/attn_old.txt
done
