#!/bin/bash
# fifth session: the default bench command (now 10 timed steps after 5 warm-up steps) end to end on the final code
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r7t; mkdir -p $O
( time timeout 900 python bench.py --strict ) > $O/bench_default.json 2> $O/bench_default.err
echo "rc=$?"; tail -c 200 $O/bench_default.json; echo; tail -4 $O/bench_default.err | cut -c1-200
python - <<'PY'
import json
s=open('gpurun_out/r7t/bench_default.json').read()
d=json.loads([l for l in s.splitlines() if l.startswith('{')][-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], d['steps'], d['warmup'], 'frac', r['frac'], r['executed_frac'], 'pmc_stale', r['pmc_stale'], 'dfwd', d['d_forward_stack']['conv_stack_frac_of_peak'], 'fid', d['fid_extract']['value'], 'failed', d['failed_legs'])
print({k:v.get('images_per_sec') for k,v in d['extra_workloads'].items()})
PY
