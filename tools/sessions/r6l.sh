#!/bin/bash
# round 6: fp32 vector pooling (bit-identical to the scalar kernel) and the FID leg in both fp32 modes after it
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r6l; mkdir -p $O
( time timeout 600 python -m pytest tests/test_eval_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -5 ) > $O/pytest_eval.txt 2>&1
grep -E " passed| failed|^FAILED|^ERROR" $O/pytest_eval.txt | head
for m in exact bf16x3; do
  timeout 300 python tools/fid_leg.py --samples 5120 --batch 256 --dtype f32 --f32-mode $m > $O/fid_leg_f32_$m.json 2> $O/fid_leg_f32_$m.err
  echo "$m: $(cut -c1-160 $O/fid_leg_f32_$m.json)"
done
timeout 300 python tools/fid_leg.py --samples 5120 --batch 256 --dtype bf16 > $O/fid_leg_bf16.json 2> $O/fid_leg_bf16.err; echo "bf16: $(cut -c1-160 $O/fid_leg_bf16.json)"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kf -o kf --output-format csv -- python $R/tools/fid_leg.py --samples 2816 --batch 256 --dtype f32 --f32-mode exact ) > $O/fid_leg_traced.json 2> $O/fid_leg_traced.err
python tools/kt_summary.py $(find $O/kf -name "*kernel_trace.csv" | head -1) 40 > $O/fid_leg_f32_exact_kerneltrace.txt 2>&1
rm -rf $O/kf
head -12 $O/fid_leg_f32_exact_kerneltrace.txt | cut -c1-170
