#!/bin/bash
# round-3 GPU session C: StyleGAN operators (f4), pipelined fused skip, mask+residual epilogue + GradLink (no autograd adds), skip microbench,
# network parity subset with everything on, step timing A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r3c
mkdir -p $O
( time timeout 300 python -m pytest tests/test_style_gpu.py -m gpu -q --maxfail=40 -p no:cacheprovider ) > $O/pytest_style.txt 2>&1
tail -12 $O/pytest_style.txt
( time timeout 300 python -m pytest tests/test_conv_v2_gpu.py tests/test_kernels_gpu.py -m gpu -q --maxfail=40 -p no:cacheprovider -k "skip or mask_and_residual or batchnorm" ) > $O/pytest_conv.txt 2>&1
tail -8 $O/pytest_conv.txt
( time timeout 200 python tools/skip_bench.py ) > $O/skip_bench.txt 2>&1
cat $O/skip_bench.txt
( time timeout 900 python -m pytest tests/test_model_gpu.py tests/test_blocks_gpu.py tests/test_fullwidth_gpu.py -m gpu -q --maxfail=20 -p no:cacheprovider -k "(biggan32 or sngan32 or wgangp32 or resgan32 or bigdeep32 or (biggan128w and golden) or (sngan32w and golden)) and not batch_curve" ) > $O/pytest_nets.txt 2>&1
tail -8 $O/pytest_nets.txt
( time timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_on.json 2> $O/bench_on.err
( time SG_GRAD_LINK=0 timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_link_off.json 2> $O/bench_link_off.err
( time SG_GRAD_LINK=0 SG_SKIP_FUSION=0 timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_both_off.json 2> $O/bench_both_off.err
python - <<'PY'
import json
for n in ("on", "link_off", "both_off"):
    try:
        d = json.load(open(f"gpurun_out/r3c/bench_{n}.json"))
        print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["conv_ms_per_step"], d["d_forward_stack"]["conv_stack_ms"], d["d_forward_stack"]["conv_launches"], d["d_forward_stack"]["forward_ms"], d["last_step_losses"])
    except Exception as e:
        print(n, "failed", e)
PY
tail -3 $O/bench_on.err
