#!/bin/bash
# round-2 GPU session H: conv_v4 (small-workgroup halo forward kernel) parity + layer table A/B, bias gradient fused into wgrad_v3
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2h
mkdir -p $O
( time timeout 900 python -m pytest tests/test_conv_v2_gpu.py -k "v4 or wgrad_v3 or v3" -m gpu -q --maxfail=40 -p no:cacheprovider --durations=5 ) > $O/pytest_gpu.txt 2>&1
tail -14 $O/pytest_gpu.txt
SEL="96-96-128,96-192-64,192-192-64,192-384-32,384-384-32,192-96-128,1536-1536-8"
SG_CONV_V4=0 timeout 300 python tools/conv_bench.py --only $SEL > $O/conv_layer_table_v4off.txt 2>&1
cat $O/conv_layer_table_v4off.txt
timeout 300 python tools/conv_bench.py --only $SEL > $O/conv_layer_table_v4.txt 2>&1
cat $O/conv_layer_table_v4.txt
SG_CONV_V4=all timeout 300 python tools/conv_bench.py --only $SEL > $O/conv_layer_table_v4all.txt 2>&1
cat $O/conv_layer_table_v4all.txt
timeout 500 python bench.py --steps 8 --warmup 3 --fid-samples 0 --no-cpu-baseline --no-extras > $O/bench_step.json 2> $O/bench_step.err
tail -c 2200 $O/bench_step.json
