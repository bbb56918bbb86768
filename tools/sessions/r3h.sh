#!/bin/bash
# round-3 GPU session H: fused filtered_lrelu, stride-2 conv_v2, 2-rank early exchange (tolerance form), FID leg with / without the stride-2 tile path
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r3h
mkdir -p $O
( time timeout 300 python -m pytest tests/test_style_gpu.py tests/test_conv_v2_gpu.py tests/test_eval_gpu.py -m gpu -q --maxfail=30 -p no:cacheprovider -k "style or filtered or bias_act or upfirdn or stride2 or inception_features" ) > $O/pytest_a.txt 2>&1
grep -E "^FAILED|passed|failed|Error" $O/pytest_a.txt | cut -c1-220 | tail -15
( time timeout 400 python -m pytest tests/test_dist_gpu.py -m gpu -q --maxfail=10 -p no:cacheprovider ) > $O/pytest_dist.txt 2>&1
grep -E "^FAILED|passed|failed" $O/pytest_dist.txt | cut -c1-220 | tail -8
( time timeout 200 python tools/fid_leg.py --samples 5120 --dtype bf16 ) > $O/fid_on.json 2> $O/fid_on.err
( time SG_CONV_V2_STRIDE2_OFF=1 timeout 200 python tools/fid_leg.py --samples 5120 --dtype bf16 ) > $O/fid_off.json 2> $O/fid_off.err
tail -1 $O/fid_on.json | cut -c1-120; tail -1 $O/fid_off.json | cut -c1-120
