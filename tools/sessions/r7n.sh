#!/bin/bash
# fifth session: ABBA-ordered step A/B of the conv_v3 tile order (SG_CONV_V3_GJ = 1 / 8) -- is there a run-order bias in the alternating A/B runs of this session?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r7n; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-extras --fid-samples 0 --no-cpu-baseline"
for gj in 1 8 8 1 1 8 8 1 1 1; do
  SG_CONV_V3_GJ=$gj timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gj=$gj', d['ms_per_step'], d['roofline']['conv_ms_per_step'], d['roofline']['per_kernel']['sg_conv_v3_kernel']['ms_per_step'], d['roofline_hbm']['batch_norm']['ms_per_step'])" | tee -a $O/abba.txt
done
