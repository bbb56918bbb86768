#!/bin/bash
# round-3 GPU session O: vectorised sg_dot + restructured projection-head backward: unit tests, network tests of the projection discriminators
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r3o
mkdir -p $O
( time timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "dot_product or projection_head" 2>&1 | tail -15 ) > $O/pytest_unit.txt 2>&1
cat $O/pytest_unit.txt
( time timeout 400 python -m pytest tests/test_blocks_gpu.py tests/test_heads_gpu.py -q -x 2>&1 | tail -8 ) > $O/pytest_nets.txt 2>&1
cat $O/pytest_nets.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt -o kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 > $GRAFT_REPO_ROOT/$O/bench_traced.json 2> $GRAFT_REPO_ROOT/$O/bench_traced.err
cd $GRAFT_REPO_ROOT
python tools/kt_summary.py $(ls $O/kt/*/*kernel_trace.csv $O/kt/*kernel_trace.csv 2>/dev/null | head -1) 130 > $O/kerneltrace.txt 2>&1
rm -rf $O/kt
grep -E "k_dot|k_pd_head|sg_conv_rs|add<|CUDAFunctor_add<c10::BFloat16|total kernel" $O/kerneltrace.txt | cut -c1-160
python - <<PY
import json
j=json.loads([l for l in open("$O/bench_traced.json") if l.startswith("{")][-1])
print("traced", j["value"], j["ms_per_step"], j["roofline"]["frac"])
PY
