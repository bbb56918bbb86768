#!/bin/bash
# round-2 GPU session B: whole GPU suite with the trimmed full-width tests + new kernels (3-buffer wgrad_v2, streaming thin-layer
# wgrad, SN / BN restructuring), layer table A/B, bench with roofline_hbm, kernel trace of the step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2b
mkdir -p $O
rm -f gpurun_out/fullwidth_parity.txt
( time timeout 1200 python -m pytest tests -m gpu -q --maxfail=80 -p no:cacheprovider --durations=25 ) > $O/pytest_gpu.txt 2>&1
tail -45 $O/pytest_gpu.txt
cp gpurun_out/fullwidth_parity.txt $O/ 2>/dev/null
timeout 300 python tools/conv_bench.py --bias > $O/conv_bench.txt 2>&1
SG_WGRAD_NBUF=2 SG_WGRAD_SK=0 timeout 300 python tools/conv_bench.py --bias > $O/conv_bench_old_wgrad.txt 2>&1
tail -26 $O/conv_bench.txt
grep -E "sum|rgb|k1" $O/conv_bench_old_wgrad.txt | tail -10
R=$PWD
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --fid-samples 0 --no-cpu-baseline ) > $O/bench_prof.json 2> $O/bench_prof.err
python tools/kt_summary.py $O/kt/kt_kernel_trace.csv 90 > $O/kerneltrace.txt 2>&1
head -60 $O/kerneltrace.txt
rm -f $O/kt/kt_kernel_trace.csv
timeout 500 python bench.py --steps 5 --warmup 2 --fid-samples 0 --no-cpu-baseline > $O/bench_quick.json 2> $O/bench_quick.err
tail -c 2500 $O/bench_quick.json
