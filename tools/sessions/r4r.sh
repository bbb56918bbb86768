#!/bin/bash
# round-4 GPU session R: the full `pytest -m gpu` suite as the driver runs it + smoke()
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4r
mkdir -p $O
rm -f gpurun_out/fullwidth_parity.txt
( time timeout 1150 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider --durations=15 ) > $O/pytest_gpu.txt 2>&1
grep -E "^FAILED|^ERROR|passed|failed|error|s call|s setup" $O/pytest_gpu.txt | cut -c1-220 | tail -45
cp gpurun_out/fullwidth_parity.txt $O/ 2>/dev/null
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.txt 2>&1
tail -4 $O/smoke.txt
