#!/bin/bash
# round-4 GPU session O: batch-norm statistics in the convolution epilogue -- kernel tests, network parity, step A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4p
mkdir -p $O
( time timeout 300 python -m pytest tests/test_quad_gpu.py -q -p no:cacheprovider --maxfail=20 2>&1 | tail -8 ) > $O/pytest_quad.txt 2>&1
cat $O/pytest_quad.txt | cut -c1-250


for f in 0 1; do
  ( SG_BN_FUSED_STATS=$f timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_f$f.json 2> $O/bench_f$f.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/bench_f$f.json") if l.startswith("{")][-1])
    r=j["roofline"]
    print("FUSED=$f", j["value"], "img/s", j["ms_per_step"], "ms; conv ms", r["conv_ms_per_step"], "bn", j["roofline_hbm"]["batch_norm"])
    for k in ("sg_conv_q_kernel", "sg_conv_v4_kernel<SKIP>"): print("  ", k, r["per_kernel"][k])
except Exception as e:
    print("failed", e)
PY
  tail -2 $O/bench_f$f.err | cut -c1-200
done
