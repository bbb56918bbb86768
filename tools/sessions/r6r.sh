#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$(pwd); O=gpurun_out/r6r; mkdir -p $O
for g in 1 4 8 16 32; do
  echo "== SG_CONV_Q_GJ=$g" >> $O/quad_gj.txt
  SG_CONV_Q_GJ=$g timeout 200 python tools/quad_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-130 >> $O/quad_gj.txt
done
cat $O/quad_gj.txt
timeout 300 python -m pytest tests/test_quad_gpu.py -x -q -m gpu 2>&1 | tail -3
SG_CONV_Q_GJ=8 timeout 300 python -m pytest tests/test_quad_gpu.py -x -q -m gpu 2>&1 | tail -3
