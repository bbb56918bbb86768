#!/bin/bash
# round-4 GPU session U (closing): default bench --strict on the final code, kernel trace + D-forward timeline of the step, PMC traffic passes,
# kernel trace of the FID leg
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r4u
mkdir -p $O
( time timeout 900 python bench.py --strict ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.json; echo; tail -3 $O/bench_default.err | cut -c1-300
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt --output-format csv -- python $R/bench.py --steps 9 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_traced.json 2> $O/bench_traced.err
KT=$(ls $O/kt/*/*kernel_trace.csv $O/kt/*kernel_trace.csv 2>/dev/null | head -1)
python tools/kt_summary.py $KT 130 > $O/kerneltrace.txt 2>&1
python tools/dfwd_timeline.py $KT > $O/dfwd_timeline.txt 2>&1
rm -rf $O/kt
head -4 $O/kerneltrace.txt | cut -c1-150; tail -3 $O/dfwd_timeline.txt
B2="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --fid-samples 0"
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/pf -o pf --output-format csv -- $B2 ) > $O/pf.log 2>&1
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$O/pw -o pw --output-format csv -- $B2 ) > $O/pw.log 2>&1
python tools/pmc_traffic.py $(ls $O/pf/*/*counter_collection.csv $O/pf/*counter_collection.csv 2>/dev/null | head -1) $(ls $O/pw/*/*counter_collection.csv $O/pw/*counter_collection.csv 2>/dev/null | head -1) > $O/conv_hbm_traffic_pmc.json 2> $O/pmc_traffic.err
head -c 400 $O/conv_hbm_traffic_pmc.json; echo; tail -2 $O/pmc_traffic.err
rm -rf $O/pf $O/pw
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kf -o kf --output-format csv -- python $R/tools/fid_leg.py --samples 2816 --batch 128 --dtype bf16 ) > $O/fid_leg.json 2> $O/fid_leg.err
python tools/kt_summary.py $(ls $O/kf/*/*kernel_trace.csv $O/kf/*kernel_trace.csv 2>/dev/null | head -1) 60 > $O/fid_leg_kerneltrace.txt 2>&1
rm -rf $O/kf
tail -c 400 $O/fid_leg.json; echo
