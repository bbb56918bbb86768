#!/bin/bash
# round-2 GPU session U: conv_v2 with partial cout tiles (InceptionV3 cout counts): parity + FID leg
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2u
mkdir -p $O
( time timeout 600 python -m pytest tests/test_conv_v2_gpu.py tests/test_eval_gpu.py -k "conv_v2_matches or inception or feature_loop" -m gpu -q --maxfail=40 -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
tail -6 $O/pytest_gpu.txt
timeout 300 python tools/fid_leg.py --samples 10240 --dtype bf16 > $O/fid_leg.json 2> $O/fid_leg.err
tail -1 $O/fid_leg.json | cut -c1-250
