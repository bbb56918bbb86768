#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$(pwd); O=gpurun_out/r6t; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "linear_group or gemm_forms" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -3
B="python bench.py --steps 10 --warmup 3 --no-extras --fid-samples 0 --no-cpu-baseline"
for rep in 1 2; do
  SG_CBN_GROUP=0 SG_CONV_Q_GJ=1 timeout 300 $B > $O/bench_base_$rep.json 2> $O/bench_base_$rep.err
  timeout 300 $B > $O/bench_new_$rep.json 2> $O/bench_new_$rep.err
  python - <<P
import json
for n in ("base","new"):
    try:
        d=json.loads(open("$O/bench_%s_$rep.json"%n).read().strip().splitlines()[-1]); print(n, $rep, d["ms_per_step"], d["value"])
    except Exception as e: print(n, "failed", e)
P
done
