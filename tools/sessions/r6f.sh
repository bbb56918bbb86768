#!/bin/bash
# round 6: the bf16 steps against the reference's golden vectors with the tightened (measured) forward bounds, test ids in the output; then bench + kernel trace of the step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r6f; mkdir -p $O
( time timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fullwidth_gpu.py -v -m gpu -p no:cacheprovider -s -k "step_vs_golden and True" 2>&1 ) > $O/pytest_bf16_steps.txt 2>&1
grep -E " passed| failed|^FAILED|^ERROR" $O/pytest_bf16_steps.txt | head
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt --output-format csv -- python $R/bench.py --steps 9 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_traced.json 2> $O/bench_traced.err
KT=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python tools/kt_summary.py $KT 160 > $O/kerneltrace.txt 2>&1
rm -rf $O/kt
head -3 $O/kerneltrace.txt | cut -c1-150; tail -c 600 $O/bench_traced.json
