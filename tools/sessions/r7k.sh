#!/bin/bash
# fifth session: batch-norm backward apply pass with non-temporal accesses on tensors beyond the Infinity Cache (same-box A/B of the step), extras with un-instrumented timed steps
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r7k; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py -q -m gpu -rf -p no:cacheprovider 2>&1 | grep -E "^FAILED|passed|failed" | tail -8
B="python bench.py --steps 10 --warmup 3 --no-extras --fid-samples 0 --no-cpu-baseline"
for z in 1 2; do
  timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('auto  ', d['ms_per_step'], d['roofline_hbm']['batch_norm'])"
  SG_BN_BWD_NT=0 timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bwd-nt0', d['ms_per_step'], d['roofline_hbm']['batch_norm'])"
done
for name in dcgan32_bs64_fp32 wgangp128_bs64_bf16 bigdeep128_bs256_bf16; do
  timeout 300 python tools/extra_run.py $name 2 2>/dev/null | tail -1 | cut -c1-700
done
