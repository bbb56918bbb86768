#!/bin/bash
# round-4 GPU session S: the test that failed in r4r again; the slow full-width comparisons; two gloo ranks on ONE device (plumbing of the
# multi-rank bench line: weak and --strong, exposed_comm_ms_per_step) -- never a scaling number
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4s
mkdir -p $O
( timeout 200 python -m pytest tests/test_blocks_gpu.py -q -p no:cacheprovider -k "gradient_penalty_double_backward" 2>&1 | tail -3 ) > $O/pytest_gp.txt 2>&1
cat $O/pytest_gp.txt
( time SG_BENCH_ONE_DEVICE=1 timeout 400 python bench.py --gpus 2 --steps 3 --warmup 2 --batch 128 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_2ranks_weak.json 2> $O/bench_2ranks_weak.err
( time SG_BENCH_ONE_DEVICE=1 timeout 400 python bench.py --gpus 2 --strong --steps 3 --warmup 2 --batch 256 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_2ranks_strong.json 2> $O/bench_2ranks_strong.err
python - <<PY
import json
for n in ("weak", "strong"):
    try:
        j=json.loads([l for l in open("$O/bench_2ranks_%s.json" % n) if l.startswith("{")][-1])
        print(n, j["value"], j["ms_per_step"], j["scaling"], j["n_gpus"], j["config"]["per_gpu_batch"], j["config"]["global_batch"], j["config"]["exchange"], "exposed", j["config"]["exposed_comm_ms_per_step"])
    except Exception as e:
        print(n, "failed", e)
PY
tail -3 $O/bench_2ranks_strong.err | cut -c1-300
bash tools/sessions/r4_slow.sh
