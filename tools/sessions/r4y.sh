#!/bin/bash
# round-4 GPU session Y: gain + bias linears of the conditional batch norms as one GEMM (CbnAffineFn) -- parity, then step A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4y
mkdir -p $O
( time timeout 500 python -m pytest tests/test_model_gpu.py tests/test_blocks_gpu.py tests/test_fullwidth_gpu.py -q -p no:cacheprovider --maxfail=10 -k "(biggan32 or bigdeep32 or sngan32 or biggan128w) and (golden or generator or G) and not stagewise" 2>&1 | tail -6 ) > $O/pytest_net.txt 2>&1
cat $O/pytest_net.txt | cut -c1-250
for f in 0 1; do
  ( SG_CBN_MERGED=$f timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_m$f.json 2> $O/bench_m$f.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/bench_m$f.json") if l.startswith("{")][-1])
    r=j["roofline"]
    print("MERGED=$f", j["value"], "img/s", j["ms_per_step"], "ms; conv ms", r["conv_ms_per_step"], "gemm ms", r["gemm_ms_per_step"], "dfwd", j["d_forward_stack"]["conv_stack_ms"], "losses", j["last_step_losses"])
except Exception as e:
    print("failed", e)
PY
  tail -2 $O/bench_m$f.err | cut -c1-200
done
