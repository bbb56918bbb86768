#!/bin/bash
# round-2 GPU session Q: conv_v4 with the 512-pixel tile (12 accumulator blocks per wave): parity + A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2q
mkdir -p $O
( time timeout 900 python -m pytest tests/test_conv_v2_gpu.py -k "conv_v4 or conv_v3" -m gpu -q --maxfail=40 -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
tail -6 $O/pytest_gpu.txt
SEL="96-96-128,96-192-64,192-192-64,192-384-32,384-384-32,192-96-128"
timeout 300 python tools/conv_bench.py --only $SEL > $O/conv_layer_table_bj256.txt 2>&1
cat $O/conv_layer_table_bj256.txt
SG_CONV_V4_BJ=512 timeout 300 python tools/conv_bench.py --only $SEL > $O/conv_layer_table_bj512.txt 2>&1
cat $O/conv_layer_table_bj512.txt
SG_CONV_V4_BJ=512 timeout 500 python bench.py --steps 8 --warmup 3 --fid-samples 0 --no-cpu-baseline --no-extras > $O/bench_step_bj512.json 2> $O/bench_step.err
tail -c 500 $O/bench_step_bj512.json
timeout 500 python bench.py --steps 8 --warmup 3 --fid-samples 0 --no-cpu-baseline --no-extras > $O/bench_step_bj256.json 2> $O/bench_step.err
tail -c 500 $O/bench_step_bj256.json
