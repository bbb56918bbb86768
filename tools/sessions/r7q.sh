#!/bin/bash
# fifth session, last call: the whole GPU suite, smoke and the default bench line on the final code (pinned table uploads; csrc unchanged since r7zz: the PMC summary stays current)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r7q; mkdir -p $O
T0=$(date +%s)
( time timeout 900 python bench.py --strict ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.json; echo; tail -4 $O/bench_default.err | cut -c1-200
( timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --fid-samples 0 --no-cpu-baseline 2>/dev/null | tail -1 ) > $O/bench_driver_flags.json
python -c "import json; d=json.loads(open('$O/bench_driver_flags.json').read()); print('20 steps / 5 warm-up:', d['ms_per_step'], d['value'], d['roofline']['pmc_stale'])"
echo "bench done at $(( $(date +%s) - T0 )) s"
( time timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=25 2>&1 | tail -60 ) > $O/pytest_gpu_full.txt 2>&1
grep -E " passed| failed|FAILED|ERROR" $O/pytest_gpu_full.txt | head -8
echo "suite done at $(( $(date +%s) - T0 )) s"
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -4 ) > $O/smoke.txt 2>&1; cat $O/smoke.txt
echo "all done at $(( $(date +%s) - T0 )) s"
