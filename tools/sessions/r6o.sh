#!/bin/bash
# round 6: bf16x3 split-precision arithmetic for fp32 TRAINING (forward, data gradient, weight gradient): parity at the exact path's tolerances, then C2 / C1 in both modes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6o; mkdir -p $O
( time timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullwidth_gpu.py -q -m gpu -p no:cacheprovider -s -k "bf16x3" 2>&1 | grep -E "weight gradient|exact|passed|failed|FAILED|FAIL|Error" | tail -40 ) > $O/pytest_split_train.txt 2>&1
cat $O/pytest_split_train.txt | cut -c1-200 | tail -25
for n in sngan32_bs256_fp32 sngan32_bs256_fp32_bf16x3; do
  timeout 300 python tools/extra_run.py $n 3 > $O/extra_$n.json 2> $O/extra_$n.err
  echo "$n: $(grep -o '"images_per_sec": [0-9.]*' $O/extra_$n.json) $(grep -o '"ms_per_step": [0-9.]*' $O/extra_$n.json) $(grep -o '"conv_ms_per_step": [0-9.]*' $O/extra_$n.json)"; tail -1 $O/extra_$n.err | cut -c1-200
done
