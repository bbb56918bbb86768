#!/bin/bash
# round-3 GPU session P: after reverting the two SN changes that measured slower in r3n (bit-identical arithmetic in every version):
# SN / network tests, default bench --strict on the final code, short kernel trace
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r3p
mkdir -p $O
( time timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x -k "spectral or sn_ or biggan32 or sngan32" 2>&1 | tail -6 ) > $O/pytest_sn.txt 2>&1
cat $O/pytest_sn.txt
( time timeout 900 python bench.py --strict ) > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/bench_default.err | cut -c1-300
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt --output-format csv -- python $R/bench.py --steps 9 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_traced.json 2> $O/bench_traced.err
python tools/kt_summary.py $(ls $O/kt/*/*kernel_trace.csv $O/kt/*kernel_trace.csv 2>/dev/null | head -1) 130 > $O/kerneltrace.txt 2>&1
rm -rf $O/kt
grep -E "total kernel|k_sn_wtu|k_sn_v$|k_sn_u$" $O/kerneltrace.txt | cut -c1-120
python - <<PY
import json
j=json.loads([l for l in open("$O/bench_default.json") if l.startswith("{")][-1])
print("default", j["value"], j["ms_per_step"], j["roofline"]["frac"], j["d_forward_stack"]["conv_stack_ms"], j["fid_extract"]["inception_bf16"], j["failed_legs"])
PY
