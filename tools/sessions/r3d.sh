#!/bin/bash
# round-3 GPU session D: GradLink with the BN-side fix, Inception KAT / manifest / sqrtm-2048, C4 at 256^2, A/B of the step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r3d
mkdir -p $O
( time timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_eval_gpu.py -m gpu -q --maxfail=40 -p no:cacheprovider -k "batchnorm or scalar_known or rejects_foreign or 2048" ) > $O/pytest_new.txt 2>&1
tail -6 $O/pytest_new.txt; cat gpurun_out/fid_backend_2048.txt
( time timeout 900 python -m pytest tests/test_model_gpu.py tests/test_blocks_gpu.py tests/test_fullwidth_gpu.py -m gpu -q --maxfail=20 -p no:cacheprovider -k "(biggan32 or sngan32 or wgangp32 or resgan32 or bigdeep32 or sngp32 or (biggan128w and golden) or (sngan32w and golden) or bigdeep256w) and not batch_curve" ) > $O/pytest_nets.txt 2>&1
grep -E "^FAILED|passed|failed" $O/pytest_nets.txt | cut -c1-160 | tail -25
( time timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_on.json 2> $O/bench_on.err
( time SG_GRAD_LINK=0 timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_link_off.json 2> $O/bench_link_off.err
( time SG_GRAD_LINK=0 SG_SKIP_FUSION=0 timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_both_off.json 2> $O/bench_both_off.err
( time timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_on2.json 2> $O/bench_on2.err
python - <<'PY'
import json
for n in ("on", "link_off", "both_off", "on2"):
    try:
        d = json.load(open(f"gpurun_out/r3d/bench_{n}.json"))
        print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["conv_ms_per_step"], d["d_forward_stack"]["conv_stack_ms"], d["d_forward_stack"]["conv_launches"], d["d_forward_stack"]["forward_ms"], d["last_step_losses"])
    except Exception as e:
        print(n, "failed", e)
PY
tail -3 $O/bench_on.err
