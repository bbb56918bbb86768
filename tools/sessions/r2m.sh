#!/bin/bash
# round-2 GPU session M: vectorised dgrad weight pack (parity through the network tests), kernel trace of the step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2m
mkdir -p $O
( time timeout 900 python -m pytest tests/test_blocks_gpu.py -k "discriminator_fwd_bwd or generator_fwd_bwd" -m gpu -q --maxfail=40 -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
tail -4 $O/pytest_gpu.txt
R=$PWD
B="python $R/bench.py --steps 3 --warmup 2 --fid-samples 0 --no-cpu-baseline --no-extras"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt --output-format csv -- $B ) > $O/bench_prof.json 2> $O/bench_prof.err
python tools/kt_summary.py $O/kt/kt_kernel_trace.csv 120 > $O/kerneltrace.txt 2>&1
head -60 $O/kerneltrace.txt
python - <<'PY' > $O/timeline_gaps.txt 2>&1
import csv, sys
rows=[]
with open('gpurun_out/r2m/kt/kt_kernel_trace.csv') as f:
    r=csv.DictReader(f)
    for x in r:
        rows.append((int(x['Start_Timestamp']), int(x['End_Timestamp'])))
rows.sort()
busy=0; gaps=0; last=rows[0][0]
for s,e in rows:
    if s>last: gaps+=s-last
    if e>last:
        busy+=e-max(s,last); last=e
print("span ms", (rows[-1][1]-rows[0][0])/1e6, "busy ms", busy/1e6, "idle ms", gaps/1e6, "launches", len(rows))
PY
cat $O/timeline_gaps.txt
rm -f $O/kt/kt_kernel_trace.csv
