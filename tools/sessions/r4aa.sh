#!/bin/bash
# round-4 GPU session AA: epilogue BN statistics in the ResNet / BigGAN-deep generators (full-width golden tests), the two extras, then the
# GPU suite without the full-width fixtures on the final code
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4aa
mkdir -p $O
( time timeout 200 python -m pytest tests/test_fullwidth_gpu.py -q -p no:cacheprovider --maxfail=5 -k "step_vs_golden and (wgangp128w or bigdeep128w)" 2>&1 | tail -3 ) > $O/pytest_wide.txt 2>&1
cat $O/pytest_wide.txt
for w in wgangp128_bs64_bf16 bigdeep128_bs256_bf16; do timeout 120 python tools/extra_run.py $w 2>/dev/null | tail -1 | cut -c1-420; done | tee $O/extras.txt
( time timeout 420 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "not fullwidth" 2>&1 | tail -4 ) > $O/pytest_rest.txt 2>&1
cat $O/pytest_rest.txt
