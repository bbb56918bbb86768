#!/bin/bash
# round 6: the bf16x3 split-precision fp32 mode -- kernel-level and Inception-level parity, then the FID leg in both fp32 modes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r6k; mkdir -p $O
( time timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_eval_gpu.py -q -m gpu -p no:cacheprovider -s -k "bf16x3 or features_at_full" 2>&1 | grep -vE "^\s*$" | tail -30 ) > $O/pytest_split.txt 2>&1
grep -E " passed| failed|^FAILED|^ERROR|exact|bf16x3" $O/pytest_split.txt | cut -c1-200 | head -30
for m in exact bf16x3; do
  timeout 300 python tools/fid_leg.py --samples 5120 --batch 256 --dtype f32 --f32-mode $m > $O/fid_leg_f32_$m.json 2> $O/fid_leg_f32_$m.err
  echo "$m: $(cut -c1-200 $O/fid_leg_f32_$m.json)"; tail -1 $O/fid_leg_f32_$m.err | cut -c1-200
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kf -o kf --output-format csv -- python $R/tools/fid_leg.py --samples 2816 --batch 256 --dtype f32 --f32-mode bf16x3 ) > $O/fid_leg_traced.json 2> $O/fid_leg_traced.err
python tools/kt_summary.py $(find $O/kf -name "*kernel_trace.csv" | head -1) 40 > $O/fid_leg_f32_bf16x3_kerneltrace.txt 2>&1
rm -rf $O/kf
head -14 $O/fid_leg_f32_bf16x3_kerneltrace.txt | cut -c1-170
