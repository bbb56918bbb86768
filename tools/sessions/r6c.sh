#!/bin/bash
# round 6: the LOGAN bounds as measured + the oracle-side tight check, APA's differentiable select, the reference-order sampler under the config-step replays
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6c; mkdir -p $O
( time timeout 900 python -m pytest tests/test_wide_info_gpu.py tests/test_wide_zz_config_steps_gpu.py tests/test_aug_gpu.py -q -m gpu -p no:cacheprovider 2>&1 ) > $O/pytest.txt 2>&1
grep -E " passed| failed|^FAILED|^ERROR" $O/pytest.txt | head -40
