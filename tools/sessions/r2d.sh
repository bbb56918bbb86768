#!/bin/bash
# round-2 GPU session D: whole GPU suite (fused attention now the default path), step bench, kernel trace, PMC HBM traffic of the conv engine
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2d
mkdir -p $O
rm -f gpurun_out/fullwidth_parity.txt
( time timeout 1200 python -m pytest tests -m gpu -q --maxfail=80 -p no:cacheprovider --durations=12 ) > $O/pytest_gpu.txt 2>&1
tail -30 $O/pytest_gpu.txt
cp gpurun_out/fullwidth_parity.txt $O/ 2>/dev/null
R=$PWD
B="python $R/bench.py --steps 3 --warmup 2 --fid-samples 0 --no-cpu-baseline --no-extras"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt --output-format csv -- $B ) > $O/bench_prof.json 2> $O/bench_prof.err
python tools/kt_summary.py $O/kt/kt_kernel_trace.csv 90 > $O/kerneltrace.txt 2>&1
head -50 $O/kerneltrace.txt
rm -f $O/kt/kt_kernel_trace.csv
B2="python $R/bench.py --steps 2 --warmup 1 --fid-samples 0 --no-cpu-baseline --no-extras"
( cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/pf -o pf --output-format csv -- $B2 ) > $O/pf.log 2>&1
( cd /tmp && timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$O/pw -o pw --output-format csv -- $B2 ) > $O/pw.log 2>&1
python tools/pmc_traffic.py $O/pf/pf_counter_collection.csv $O/pw/pw_counter_collection.csv > $O/conv_hbm_traffic_pmc.json 2> $O/pmc_traffic.err
head -8 $O/conv_hbm_traffic_pmc.json
rm -f $O/pf/pf_kernel_trace.csv $O/pw/pw_kernel_trace.csv
timeout 500 python bench.py --steps 10 --warmup 3 --fid-samples 0 --no-cpu-baseline --no-extras > $O/bench_step.json 2> $O/bench_step.err
tail -c 2500 $O/bench_step.json
