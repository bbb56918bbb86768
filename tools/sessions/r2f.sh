#!/bin/bash
# round-2 GPU session F: flash-style attention forward + single-pass query-side backward, halo weight gradient down to 8 x 8 images
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2f
mkdir -p $O
( time timeout 900 python -m pytest tests/test_conv_v2_gpu.py tests/test_kernels_gpu.py tests/test_blocks_gpu.py "tests/test_model_gpu.py::test_training_step_vs_golden" \
    -k "wgrad_v3 or attention or attn or biggan or bigdeep or big" -m gpu -q --maxfail=40 -p no:cacheprovider --durations=8 ) > $O/pytest_gpu.txt 2>&1
tail -30 $O/pytest_gpu.txt
timeout 300 python tools/conv_bench.py > $O/conv_layer_table.txt 2> $O/conv_layer_table.err
cat $O/conv_layer_table.txt
R=$PWD
B="python $R/bench.py --steps 3 --warmup 2 --fid-samples 0 --no-cpu-baseline --no-extras"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt --output-format csv -- $B ) > $O/bench_prof.json 2> $O/bench_prof.err
python tools/kt_summary.py $O/kt/kt_kernel_trace.csv 70 > $O/kerneltrace.txt 2>&1
head -45 $O/kerneltrace.txt
rm -f $O/kt/kt_kernel_trace.csv
timeout 500 python bench.py --steps 8 --warmup 3 --fid-samples 0 --no-cpu-baseline --no-extras > $O/bench_step.json 2> $O/bench_step.err
tail -c 2500 $O/bench_step.json
