#!/bin/bash
# ablation of conv_v4.h's costs on the shallow 3x3 layers (SG_V4_ABLATE bit mask: 1 no epilogue, 2 no patch reload at slice boundaries, 4 no weight DMA in the loop,
# 8 no barrier per tap, 16 no fragment reads, 32 epilogue without its global stores)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6u; mkdir -p $O; rm -f $O/v4_ablation.txt
for a in 0 32 1 2 4 7 15 31 0; do
  echo "== SG_V4_ABLATE=$a" >> $O/v4_ablation.txt
  SG_V4_ABLATE=$a timeout 200 python tools/conv_bench.py --only 96-192-64,192-192-64,192-384-32,384-384-32 2>&1 | grep -v "amdgpu.ids\|k1 " | cut -c1-100 >> $O/v4_ablation.txt
done
cat $O/v4_ablation.txt
