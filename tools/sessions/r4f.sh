#!/bin/bash
# round-4 GPU session F: batched quad packing, 128-pixel tile, fused skip in the POOL-form quad kernel, rs96 default for the plain variant
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r4h
mkdir -p $O
( time timeout 300 python -m pytest tests/test_quad_gpu.py -q -p no:cacheprovider --maxfail=20 2>&1 | tail -8 ) > $O/pytest_quad.txt 2>&1
cat $O/pytest_quad.txt | cut -c1-250
( time timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench.json 2> $O/bench.err
python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
    print(j["value"], "img/s", j["ms_per_step"], "ms; conv frac", j["roofline"]["frac"], "conv ms", j["roofline"]["conv_ms_per_step"], "dfwd", j["d_forward_stack"], "losses", j["last_step_losses"])
    print({k: v for k, v in j["roofline_hbm"].items() if isinstance(v, dict)})
except Exception as e:
    print("failed", e)
PY
tail -3 $O/bench.err | cut -c1-300
timeout 200 python tools/quad_bench.py > $O/quad_bench.txt 2>&1; cat $O/quad_bench.txt
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_traced.json 2> $O/bench_traced.err
KT=$(ls $O/kt/*/*kernel_trace.csv $O/kt/*kernel_trace.csv 2>/dev/null | head -1)
python tools/kt_summary.py $KT 130 > $O/kerneltrace.txt 2>&1
python tools/dfwd_timeline.py $KT > $O/dfwd_timeline.txt 2>&1
rm -rf $O/kt
tail -36 $O/dfwd_timeline.txt
head -45 $O/kerneltrace.txt | cut -c1-150
