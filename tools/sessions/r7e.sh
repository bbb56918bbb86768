#!/bin/bash
# fifth session: same-box A/B of the timed region's instrumentation: event pairs around the dominant kernel only (default) vs around every engine launch (SG_BENCH_PROF_ALL=1, the
# arrangement until now); traced run of the default arrangement for the gap table
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r7e; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-extras --fid-samples 0 --no-cpu-baseline"
for z in 1 2; do
  timeout 300 $B 2>$O/err_dom_$z.txt | tail -1 > $O/bench_dom_$z.json
  python -c "import sys,json; d=json.loads(open('$O/bench_dom_$z.json').read()); r=d['roofline']; print('dominant-only', d['ms_per_step'], 'family', r['frac'], r['executed_frac'], r['conv_ms_per_step'], 'dom', r['dominant_kernel'])" || tail -5 $O/err_dom_$z.txt
  SG_BENCH_PROF_ALL=1 timeout 300 $B 2>$O/err_all_$z.txt | tail -1 > $O/bench_all_$z.json
  python -c "import sys,json; d=json.loads(open('$O/bench_all_$z.json').read()); r=d['roofline']; print('all-launches ', d['ms_per_step'], 'family', r['frac'], r['executed_frac'], r['conv_ms_per_step'], 'dom', r['dominant_kernel'])" || tail -5 $O/err_all_$z.txt
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt --output-format csv -- python $R/bench.py --steps 9 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_traced.json 2> $O/bench_traced.err
KT=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python tools/kt_gaps.py $KT 6 3 2 > $O/gaps.txt 2>&1
rm -rf $O/kt
head -12 $O/gaps.txt | cut -c1-200
