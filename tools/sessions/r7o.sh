#!/bin/bash
# fifth session: where the host spends the first steps after a synchronisation (tools/host_probe.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r7o; mkdir -p $O
for m in default; do timeout 300 python tools/host_probe.py --steps 6 --warmup 3 --gc $m 2>&1 | grep -v amdgpu.ids | grep -E 'Tensor.to|: host|lead' | cut -c1-260 | tee -a $O/host_probe_gc.txt; done
