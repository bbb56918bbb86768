#!/bin/bash
# round-3 GPU session A: live-discriminator bench baseline, bf16 parity at the benchmarked widths (4 nets + batch curve), image-row-parity swizzle A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r3a
mkdir -p $O
rm -f gpurun_out/fullwidth_parity.txt
( time timeout 420 python -m pytest tests/test_conv_v2_gpu.py tests/test_heads_gpu.py tests/test_dist_gpu.py tests/test_eval_gpu.py -m gpu -q --maxfail=40 -p no:cacheprovider -k "not pil and not sqrtm" ) > $O/pytest_kernels.txt 2>&1
tail -5 $O/pytest_kernels.txt
( time timeout 300 python tools/conv_bench.py --bias ) > $O/layer_table_swz1.txt 2>&1
( time SG_SWZ_PAR=0 timeout 300 python tools/conv_bench.py --bias ) > $O/layer_table_swz0.txt 2>&1
tail -2 $O/layer_table_swz1.txt $O/layer_table_swz0.txt
( time timeout 900 python -m pytest tests/test_fullwidth_gpu.py -m gpu -q --maxfail=40 -p no:cacheprovider --durations=6 ) > $O/pytest_fullwidth.txt 2>&1
tail -12 $O/pytest_fullwidth.txt
cp gpurun_out/fullwidth_parity.txt $O/ 2>/dev/null
( time timeout 700 python bench.py --strict ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 2500 $O/bench_default.json; tail -5 $O/bench_default.err
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_traced.json 2> $O/bench_traced.err
python tools/kt_summary.py $(ls $O/kt/*/*kernel_trace.csv $O/kt/*kernel_trace.csv 2>/dev/null | head -1) 120 > $O/kerneltrace.txt 2>&1
head -30 $O/kerneltrace.txt
rm -rf $O/kt
