#!/bin/bash
# fifth session: batch-norm apply pass walking its map back to front (SG_BN_APPLY_REV=1: its last stores cover the start of the map, where the convolution behind it begins to read), ABBA step A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r7r; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -k "batchnorm or bn" 2>&1 | grep -E "passed|failed" | tail -2
SG_BN_APPLY_REV=1 timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -k "batchnorm or bn" 2>&1 | grep -E "passed|failed" | tail -2
B="python bench.py --steps 12 --warmup 4 --no-extras --fid-samples 0 --no-cpu-baseline"
for rev in 0 1 1 0 0 1 1 0; do
  SG_BN_APPLY_REV=$rev timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rev=$rev', d['ms_per_step'], d['roofline']['conv_ms_per_step'], d['roofline_hbm']['batch_norm']['ms_per_step'])" | tee -a $O/abba.txt
done
