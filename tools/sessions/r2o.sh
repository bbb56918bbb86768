#!/bin/bash
# round-2 GPU session O: vector pooling kernel (parity + FID leg), kernel traces of the BigGAN-deep and WGAN-GP extra workloads
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2o
mkdir -p $O
( time timeout 600 python -m pytest tests/test_eval_gpu.py -k "pool2d or inception" -m gpu -q --maxfail=40 -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
timeout 300 python tools/fid_leg.py --samples 10240 --dtype bf16 > $O/fid_leg.json 2> $O/fid_leg.err
tail -1 $O/fid_leg.json
R=$PWD
for W in bigdeep128_bs256_bf16 wgangp128_bs64_bf16; do
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/kt_$W -o kt --output-format csv -- python $R/tools/extra_run.py $W 2 ) > $O/extra_$W.json 2> $O/extra_$W.err
  python tools/kt_summary.py $O/kt_$W/kt_kernel_trace.csv 60 > $O/kerneltrace_$W.txt 2>&1
  rm -f $O/kt_$W/kt_kernel_trace.csv
  tail -1 $O/extra_$W.json | cut -c1-400
  head -45 $O/kerneltrace_$W.txt
done
