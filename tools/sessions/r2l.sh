#!/bin/bash
# round-2 GPU session L: four-wave (one wave per SIMD) variants of the deep-layer halo tiles: parity + A/B on the deep layers + step A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2l
mkdir -p $O
( time timeout 900 python -m pytest tests/test_conv_v2_gpu.py -k "conv_v3" -m gpu -q --maxfail=40 -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
SEL="384-768-16,768-768-16,768-1536-8,1536-1536-8,1536-768-16,1536-1536-4"
timeout 300 python tools/conv_bench.py --only $SEL > $O/conv_layer_table_nw8.txt 2>&1
cat $O/conv_layer_table_nw8.txt
SG_V3_NW4=1 timeout 300 python tools/conv_bench.py --only $SEL > $O/conv_layer_table_nw4.txt 2>&1
cat $O/conv_layer_table_nw4.txt
SG_V3_NW4=1 timeout 500 python bench.py --steps 8 --warmup 3 --fid-samples 0 --no-cpu-baseline --no-extras > $O/bench_step_nw4.json 2> $O/bench_step.err
tail -c 600 $O/bench_step_nw4.json
