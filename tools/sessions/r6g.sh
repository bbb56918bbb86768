#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6g; mkdir -p $O
timeout 400 python tools/probes/rccl_two_ranks_one_device.py > $O/rccl_probe.txt 2>&1; grep -E "^torch-nccl|^native" $O/rccl_probe.txt | cut -c1-400
