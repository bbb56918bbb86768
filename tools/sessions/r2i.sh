#!/bin/bash
# round-2 GPU session I: conv_v4 with weights two taps ahead (counted vmcnt), dispatch rule C <= 384; whole-table A/B; step bench + kernel trace
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2i
mkdir -p $O
( time timeout 900 python -m pytest tests/test_conv_v2_gpu.py -k "v4" -m gpu -q --maxfail=40 -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
timeout 300 python tools/conv_bench.py > $O/conv_layer_table.txt 2>&1
cat $O/conv_layer_table.txt
SG_CONV_V4=all timeout 300 python tools/conv_bench.py --only "384-384-32,384-768-16,768-768-16,768-1536-8,1536-1536-8,1536-768-16,1536-1536-4" > $O/conv_layer_table_v4all.txt 2>&1
cat $O/conv_layer_table_v4all.txt
timeout 500 python bench.py --steps 8 --warmup 3 --fid-samples 0 --no-cpu-baseline --no-extras > $O/bench_step.json 2> $O/bench_step.err
tail -c 2200 $O/bench_step.json
