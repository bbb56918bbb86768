#!/bin/bash
# round-4 GPU session D: the quad kernels inside the networks -- network-level parity tests, then the step A/B (SG_QUAD=0 vs 1) on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4d
mkdir -p $O
for q in 0 1; do
  ( time SG_QUAD=$q timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_quad$q.json 2> $O/bench_quad$q.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/bench_quad$q.json") if l.startswith("{")][-1])
    print("SG_QUAD=$q", j["value"], "img/s", j["ms_per_step"], "ms; conv frac", j["roofline"]["frac"], "conv ms", j["roofline"]["conv_ms_per_step"], "dfwd", j["d_forward_stack"]["conv_stack_ms"], j["d_forward_stack"]["conv_stack_frac_of_peak"], "losses", j["last_step_losses"])
except Exception as e:
    print("SG_QUAD=$q failed", e)
PY
  tail -3 $O/bench_quad$q.err | cut -c1-300
done
( time timeout 900 python -m pytest tests/test_model_gpu.py tests/test_blocks_gpu.py tests/test_fullwidth_gpu.py -q -p no:cacheprovider --maxfail=12 --durations=12 -k "not batch_curve and not bigdeep256w" 2>&1 | tail -45 ) > $O/pytest_net.txt 2>&1
cat $O/pytest_net.txt | cut -c1-220
