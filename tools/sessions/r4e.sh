#!/bin/bash
# round-4 GPU session E: kernel table of the step and per-launch D-forward timeline with the quad kernels on
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r4e
mkdir -p $O
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_traced.json 2> $O/bench_traced.err
KT=$(ls $O/kt/*/*kernel_trace.csv $O/kt/*kernel_trace.csv 2>/dev/null | head -1)
python tools/kt_summary.py $KT 130 > $O/kerneltrace.txt 2>&1
python tools/dfwd_timeline.py $KT > $O/dfwd_timeline.txt 2>&1
rm -rf $O/kt
tail -45 $O/dfwd_timeline.txt
head -60 $O/kerneltrace.txt | cut -c1-160
