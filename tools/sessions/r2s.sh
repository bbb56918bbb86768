#!/bin/bash
# round-2 GPU session S (closing): whole GPU suite, smoke(), default bench -- on the final code state
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2s
mkdir -p $O
rm -f gpurun_out/fullwidth_parity.txt
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=80 -p no:cacheprovider --durations=8 ) > $O/pytest_gpu.txt 2>&1
tail -16 $O/pytest_gpu.txt
cp gpurun_out/fullwidth_parity.txt $O/ 2>/dev/null
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.txt 2>&1
tail -4 $O/smoke.txt
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 1500 $O/bench_default.json
