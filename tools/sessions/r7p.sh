#!/bin/bash
# fifth session: descriptor tables uploaded from pinned memory without blocking the host (L.upload_bytes): host probe, model tests, default bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r7p; mkdir -p $O
timeout 300 python tools/host_probe.py --steps 6 --warmup 3 2>&1 | grep -v amdgpu.ids | grep -E 'Tensor.to|: host|lead' | cut -c1-200 | tee $O/host_probe.txt
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_blocks_gpu.py tests/test_sn_gpu.py -q -m gpu -rf -p no:cacheprovider 2>&1 | grep -E "^FAILED|passed|failed" | tail -5
for z in 1 2; do
  timeout 300 python bench.py --no-extras --fid-samples 0 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default (5 steps, 2 warm-up)', d['ms_per_step'], d['host_lead']['gpu_minus_host_ms_per_step_end'])"
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --fid-samples 0 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver flags (20 steps, 5 warm-up)', d['ms_per_step'], d['host_lead']['gpu_minus_host_ms_per_step_end'][:8])"
