#!/bin/bash
# fifth session: step-level A/B of the batch-norm forward apply policy (non-temporal + 4096 workgroups on maps beyond the Infinity Cache) against plain accesses, three alternations
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r7l; mkdir -p $O
B="python bench.py --steps 12 --warmup 3 --no-extras --fid-samples 0 --no-cpu-baseline"
for z in 1 2 3; do
  timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('auto(nt)', d['ms_per_step'], d['roofline_hbm']['batch_norm']['ms_per_step'], d['roofline']['conv_ms_per_step'])"
  SG_BN_APPLY=02 timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain   ', d['ms_per_step'], d['roofline_hbm']['batch_norm']['ms_per_step'], d['roofline']['conv_ms_per_step'])"
done
