#!/bin/bash
# round-3 GPU session B: fused 3x3 + 1x1-skip launch (conv_v4 SKIP): kernel test, network parity with it on, step / D-forward timing on vs off
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r3b
mkdir -p $O
rm -f gpurun_out/fullwidth_parity.txt
( time timeout 300 python -m pytest tests/test_conv_v2_gpu.py -m gpu -q --maxfail=40 -p no:cacheprovider -k "skip" ) > $O/pytest_skip.txt 2>&1
tail -15 $O/pytest_skip.txt
( time timeout 900 python -m pytest tests/test_fullwidth_gpu.py -m gpu -q --maxfail=40 -p no:cacheprovider --durations=6 ) > $O/pytest_fullwidth.txt 2>&1
tail -12 $O/pytest_fullwidth.txt
cp gpurun_out/fullwidth_parity.txt $O/ 2>/dev/null
( time timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_skip_on.json 2> $O/bench_skip_on.err
( time SG_SKIP_FUSION=0 timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_skip_off.json 2> $O/bench_skip_off.err
python - <<'PY'
import json
for n in ("on", "off"):
    try:
        d = json.load(open(f"gpurun_out/r3b/bench_skip_{n}.json"))
        print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["conv_ms_per_step"], d["d_forward_stack"]["conv_stack_ms"], d["d_forward_stack"]["conv_launches"], d["d_forward_stack"]["forward_ms"], d["last_step_losses"])
    except Exception as e:
        print(n, "failed", e)
PY
tail -3 $O/bench_skip_on.err
( time timeout 600 python -m pytest tests/test_model_gpu.py tests/test_blocks_gpu.py -m gpu -q --maxfail=20 -p no:cacheprovider -k "biggan32 or sngan32 or wgangp32 or resgan32" ) > $O/pytest_nets.txt 2>&1
tail -5 $O/pytest_nets.txt
