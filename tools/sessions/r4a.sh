#!/bin/bash
# round-4 GPU session A: first execution of conv_rs96.h (parity vs CPU fp64 + halo kernel), its layer timing against the halo kernel,
# and the baseline per-launch D-forward timeline + kernel table of the step on this round's starting code
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r4a
mkdir -p $O
( time SG_EXPERIMENTAL=1 timeout 240 python -m pytest tests/test_conv_v2_gpu.py -q -x -k "rs96" -p no:cacheprovider 2>&1 | tail -15 ) > $O/pytest_rs96.txt 2>&1
cat $O/pytest_rs96.txt
for m in 0 1; do
  echo "== SG_CONV_RS96=$m" >> $O/conv_bench_rs96.txt
  SG_CONV_RS96=$m timeout 200 python tools/conv_bench.py --bias --only 96-96-128 >> $O/conv_bench_rs96.txt 2>&1
done
cat $O/conv_bench_rs96.txt
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_traced.json 2> $O/bench_traced.err
KT=$(ls $O/kt/*/*kernel_trace.csv $O/kt/*kernel_trace.csv 2>/dev/null | head -1)
python tools/kt_summary.py $KT 130 > $O/kerneltrace.txt 2>&1
python tools/dfwd_timeline.py $KT > $O/dfwd_timeline.txt 2>&1
rm -rf $O/kt
tail -40 $O/dfwd_timeline.txt
head -30 $O/kerneltrace.txt | cut -c1-160
