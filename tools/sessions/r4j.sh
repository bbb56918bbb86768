#!/bin/bash
# round-4 GPU session J: teacher-forced generator comparisons with the measured floor; bench line with per-kernel table
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4j
mkdir -p $O
rm -f gpurun_out/fullwidth_parity.txt
( time timeout 600 python -m pytest tests/test_blocks_gpu.py tests/test_fullwidth_gpu.py -q -p no:cacheprovider --maxfail=30 -k "bf16_vs_emulating and (biggan or bigdeep) and G" 2>&1 | grep -E "teacher|FAIL|passed|failed|Error|error" | cut -c1-230 ) > $O/pytest_teacher.txt 2>&1
grep -E "FAIL|passed|failed|rror|teacher-forced:" $O/pytest_teacher.txt | head -60
( time timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench.json 2> $O/bench.err
python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
    r=j["roofline"]
    print(j["value"], "img/s", j["ms_per_step"], "ms; frac", r["frac"], "exec", r["executed_frac"], "dominant", r["dominant_kernel"], "bytes/launch", r["algorithmic_bytes_per_launch"])
    for k,v in r["per_kernel"].items(): print("  ", k, v)
    d=j["d_forward_stack"]; print({k:v for k,v in d.items() if k!="per_kernel"})
    for k,v in d["per_kernel"].items(): print("  ", k, v)
except Exception as e:
    print("failed", e)
PY
tail -3 $O/bench.err | cut -c1-300
