#!/bin/bash
# round-3 GPU session M: conv_rs with the DMA pieces behind the first MFMAs: parity + microbench (warm: the pair is run twice) + step A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r3m
mkdir -p $O
( time timeout 300 python -m pytest tests/test_conv_v2_gpu.py -q -k "conv_rs" 2>&1 | tail -15 ) > $O/pytest_rs.txt 2>&1
cat $O/pytest_rs.txt
for m in 1 0; do
  echo "== SG_CONV_RS=$m" >> $O/conv_bench_rs.txt
  SG_CONV_RS=$m timeout 120 python tools/conv_bench.py --only 96-96-128,96-8-128,8-96-128 --bias 2>&1 | grep -v amdgpu.ids >> $O/conv_bench_rs.txt
done
cat $O/conv_bench_rs.txt
for m in 1 0; do
  SG_CONV_RS=$m timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --fid-samples 0 > $O/bench_rs$m.json 2> $O/bench_rs$m.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/bench_rs$m.json") if l.startswith("{")][-1])
    print("rs=$m", j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["launches_per_step"], j["last_step_losses"])
except Exception as e:
    print("rs=$m failed", e); print(open("$O/bench_rs$m.err").read()[-1500:])
PY
done
