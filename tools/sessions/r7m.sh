#!/bin/bash
# fifth session: weight-stationary tile order in conv_v3 (SG_CONV_V3_GJ = pixel tiles per group): layer table per value, bit-identity test, step A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r7m; mkdir -p $O
for gj in 1 4 8 16; do
  echo "== SG_CONV_V3_GJ=$gj"
  SG_CONV_V3_GJ=$gj timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | grep -E "k3 *$|k3 up|shape|sum" | cut -c1-120 | tee -a $O/conv_bench_gj.txt
done
SG_CONV_V3_GJ=8 timeout 600 python -m pytest tests/test_conv_v2_gpu.py tests/test_kernels_gpu.py -q -m gpu -rf -p no:cacheprovider -k "conv" 2>&1 | grep -E "^FAILED|passed|failed" | tail -5
B="python bench.py --steps 10 --warmup 3 --no-extras --fid-samples 0 --no-cpu-baseline"
for z in 1 2; do
  for gj in 1 8; do
    SG_CONV_V3_GJ=$gj timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gj=$gj', d['ms_per_step'], d['roofline']['conv_ms_per_step'], d['roofline']['per_kernel']['sg_conv_v3_kernel']['ms_per_step'])"
  done
done
