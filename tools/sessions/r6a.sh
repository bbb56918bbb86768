#!/bin/bash
# round-6 opening GPU session: the tests the driver's -x run never reached in r05 (and the one that failed), without -x, full output
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6a
mkdir -p $O
T0=$(date +%s)
( time timeout 900 python -m pytest tests/test_wide_info_gpu.py -q -m gpu -p no:cacheprovider -k "logan or r1_with or preparation" 2>&1 ) > $O/pytest_wide_info.txt 2>&1
grep -E " passed| failed|^FAILED|^ERROR" $O/pytest_wide_info.txt | head -40
echo "wide_info done at $(( $(date +%s) - T0 )) s"
( time timeout 1200 python -m pytest tests/test_wide_zz_config_steps_gpu.py -q -m gpu -p no:cacheprovider 2>&1 ) > $O/pytest_config_steps.txt 2>&1
grep -E " passed| failed|^FAILED|^ERROR" $O/pytest_config_steps.txt | head -40
echo "config_steps done at $(( $(date +%s) - T0 )) s"
