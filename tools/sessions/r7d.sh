#!/bin/bash
# fifth session: host lead of the untraced C3 step (bench.py host_lead) and idle gaps of the traced TIMED steps (tools/kt_gaps.py, the two profiled steps at the end left out)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r7d; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-extras --fid-samples 0 --no-cpu-baseline"
timeout 300 $B 2>/dev/null | tail -1 > $O/bench_plain.json
python -c "import sys,json; d=json.loads(open('$O/bench_plain.json').read()); print('c3', d['ms_per_step'], d['host_lead'])"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt --output-format csv -- python $R/bench.py --steps 9 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_traced.json 2> $O/bench_traced.err
KT=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python tools/kt_gaps.py $KT 6 3 2 > $O/gaps.txt 2>&1
rm -rf $O/kt
cat $O/gaps.txt | cut -c1-200
tail -1 $O/bench_traced.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('traced c3', d['ms_per_step'], d['host_lead'])"
