#!/bin/bash
# round-3 GPU session E: early gradient exchange (2 ranks on one GPU), 16-byte split-K reduce A/B, shared bias gradient, traces of the step and of the FID leg
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r3e
mkdir -p $O
( time timeout 600 python -m pytest tests/test_dist_gpu.py -m gpu -q --maxfail=10 -p no:cacheprovider ) > $O/pytest_dist.txt 2>&1
grep -E "^FAILED|passed|failed|Error" $O/pytest_dist.txt | cut -c1-200 | tail -12
( time timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_conv_v2_gpu.py tests/test_model_gpu.py -m gpu -q --maxfail=20 -p no:cacheprovider -k "wgrad or (biggan32 and golden) or (sngan32 and golden) or skip" ) > $O/pytest_wgrad.txt 2>&1
grep -E "^FAILED|passed|failed" $O/pytest_wgrad.txt | cut -c1-200 | tail -8
( time timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_on.json 2> $O/bench_on.err
( time SG_REDUCE_V4=0 timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_reduce_scalar.json 2> $O/bench_reduce_scalar.err
python - <<'PY'
import json
for n in ("on", "reduce_scalar"):
    try:
        d = json.load(open(f"gpurun_out/r3e/bench_{n}.json"))
        print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["conv_ms_per_step"], d["d_forward_stack"]["conv_stack_ms"], d["last_step_losses"])
    except Exception as e:
        print(n, "failed", e)
PY
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_traced.json 2> $O/bench_traced.err
python tools/kt_summary.py $(ls $O/kt/*/*kernel_trace.csv $O/kt/*kernel_trace.csv 2>/dev/null | head -1) 120 > $O/kerneltrace.txt 2>&1
head -12 $O/kerneltrace.txt | cut -c1-150
rm -rf $O/kt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kf -o kf --output-format csv -- python $R/tools/fid_leg.py --samples 5120 --dtype bf16 ) > $O/fid_leg.json 2> $O/fid_leg.err
python tools/kt_summary.py $(ls $O/kf/*/*kernel_trace.csv $O/kf/*kernel_trace.csv 2>/dev/null | head -1) 60 > $O/fid_leg_kerneltrace.txt 2>&1
head -14 $O/fid_leg_kerneltrace.txt | cut -c1-150; cat $O/fid_leg.json | tail -1 | cut -c1-300
rm -rf $O/kf
