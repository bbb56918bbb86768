#!/bin/bash
# round-4 GPU session L: double-buffered conv_q variant (parity + A/B), batched pack kernel v2, step A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4l
mkdir -p $O
( time timeout 300 python -m pytest tests/test_quad_gpu.py -q -p no:cacheprovider --maxfail=20 2>&1 | tail -8 ) > $O/pytest_quad.txt 2>&1
cat $O/pytest_quad.txt | cut -c1-250
for db in 0 1; do
  echo "== SG_CONV_Q_DB=$db" >> $O/quad_bench.txt
  SG_CONV_Q_DB=$db timeout 200 python tools/quad_bench.py >> $O/quad_bench.txt 2>&1
done
cat $O/quad_bench.txt | cut -c1-110
for db in 0 1; do
  ( SG_CONV_Q_DB=$db timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_db$db.json 2> $O/bench_db$db.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/bench_db$db.json") if l.startswith("{")][-1])
    r=j["roofline"]
    print("DB=$db", j["value"], "img/s", j["ms_per_step"], "ms; frac", r["frac"], "conv ms", r["conv_ms_per_step"], "dfwd", j["d_forward_stack"]["conv_stack_ms"], j["d_forward_stack"]["conv_stack_frac_of_peak"])
    for k in ("sg_conv_q_kernel", "sg_conv_q_kernel<SKIP>"): print("  ", k, r["per_kernel"][k])
except Exception as e:
    print("failed", e)
PY
done
