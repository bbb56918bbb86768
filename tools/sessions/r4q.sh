#!/bin/bash
# round-4 GPU session Q: image skip fused into the first discriminator block's quad launch
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4q
mkdir -p $O
( time timeout 300 python -m pytest tests/test_quad_gpu.py -q -p no:cacheprovider --maxfail=20 2>&1 | tail -8 ) > $O/pytest_quad.txt 2>&1
cat $O/pytest_quad.txt | cut -c1-250
( time timeout 600 python -m pytest tests/test_model_gpu.py tests/test_blocks_gpu.py tests/test_fullwidth_gpu.py -q -p no:cacheprovider --maxfail=10 -k "(biggan32 or biggan128w) and (golden or step or discriminator or D)" 2>&1 | tail -8 ) > $O/pytest_net.txt 2>&1
cat $O/pytest_net.txt | cut -c1-250
( timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench.json 2> $O/bench.err
python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
    r=j["roofline"]
    print(j["value"], "img/s", j["ms_per_step"], "ms; frac", r["frac"], "conv ms", r["conv_ms_per_step"], {k:v for k,v in j["d_forward_stack"].items() if k!="per_kernel"})
except Exception as e:
    print("failed", e)
PY
tail -2 $O/bench.err | cut -c1-200
