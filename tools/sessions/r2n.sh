#!/bin/bash
# round-2 GPU session N (final): whole GPU suite, default bench (extras, FID leg, CPU baseline), kernel trace, PMC HBM traffic of the conv engine, FID-leg trace
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2n
mkdir -p $O
rm -f gpurun_out/fullwidth_parity.txt
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=80 -p no:cacheprovider --durations=12 ) > $O/pytest_gpu.txt 2>&1
tail -25 $O/pytest_gpu.txt
cp gpurun_out/fullwidth_parity.txt $O/ 2>/dev/null
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/bench_default.err
tail -c 3000 $O/bench_default.json
R=$PWD
B="python $R/bench.py --steps 3 --warmup 2 --fid-samples 0 --no-cpu-baseline --no-extras"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt --output-format csv -- $B ) > $O/bench_prof.json 2> $O/bench_prof.err
python tools/kt_summary.py $O/kt/kt_kernel_trace.csv 120 > $O/kerneltrace.txt 2>&1
head -30 $O/kerneltrace.txt
rm -f $O/kt/kt_kernel_trace.csv
B2="python $R/bench.py --steps 2 --warmup 1 --fid-samples 0 --no-cpu-baseline --no-extras"
( cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/pf -o pf --output-format csv -- $B2 ) > $O/pf.log 2>&1
( cd /tmp && timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$O/pw -o pw --output-format csv -- $B2 ) > $O/pw.log 2>&1
python tools/pmc_traffic.py $O/pf/pf_counter_collection.csv $O/pw/pw_counter_collection.csv > $O/conv_hbm_traffic_pmc.json 2> $O/pmc_traffic.err
head -8 $O/conv_hbm_traffic_pmc.json
rm -f $O/pf/pf_kernel_trace.csv $O/pw/pw_kernel_trace.csv $O/pf/pf_counter_collection.csv $O/pw/pw_counter_collection.csv
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kf -o kf --output-format csv -- python $R/tools/fid_leg.py --samples 5120 --dtype bf16 ) > $O/fid_leg.json 2> $O/fid_leg.err
python tools/kt_summary.py $O/kf/kf_kernel_trace.csv 60 > $O/fid_leg_kerneltrace.txt 2>&1
tail -2 $O/fid_leg.json; head -25 $O/fid_leg_kerneltrace.txt
rm -f $O/kf/kf_kernel_trace.csv
