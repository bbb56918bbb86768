#!/bin/bash
# round-3 GPU session N (closing, second): GPU suite minus the batch-curve / 256^2 fixtures (unchanged code paths, 7 min of CPU-oracle time),
# smoke(), default bench --strict, PMC traffic passes, kernel traces of the step and of the FID leg -- on the final code state
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r3n
mkdir -p $O
rm -f gpurun_out/fullwidth_parity.txt
( time timeout 900 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider --durations=6 -k "not batch_curve and not bigdeep256w" ) > $O/pytest_gpu.txt 2>&1
grep -E "^FAILED|passed|failed|error" $O/pytest_gpu.txt | cut -c1-200 | tail -20
cp gpurun_out/fullwidth_parity.txt $O/ 2>/dev/null
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.txt 2>&1
tail -3 $O/smoke.txt
( time timeout 900 python bench.py --strict ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 1000 $O/bench_default.json; echo; tail -3 $O/bench_default.err | cut -c1-300
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt --output-format csv -- python $R/bench.py --steps 9 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_traced.json 2> $O/bench_traced.err
python tools/kt_summary.py $(ls $O/kt/*/*kernel_trace.csv $O/kt/*kernel_trace.csv 2>/dev/null | head -1) 130 > $O/kerneltrace.txt 2>&1
rm -rf $O/kt
head -4 $O/kerneltrace.txt | cut -c1-150
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kf -o kf --output-format csv -- python $R/tools/fid_leg.py --samples 2816 --batch 128 --dtype bf16 ) > $O/fid_leg.json 2> $O/fid_leg.err
python tools/kt_summary.py $(ls $O/kf/*/*kernel_trace.csv $O/kf/*kernel_trace.csv 2>/dev/null | head -1) 60 > $O/fid_leg_kerneltrace.txt 2>&1
rm -rf $O/kf
tail -c 600 $O/fid_leg.json; echo
B2="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --fid-samples 0"
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/pf -o pf --output-format csv -- $B2 ) > $O/pf.log 2>&1
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$O/pw -o pw --output-format csv -- $B2 ) > $O/pw.log 2>&1
python tools/pmc_traffic.py $(ls $O/pf/*/*counter_collection.csv $O/pf/*counter_collection.csv 2>/dev/null | head -1) $(ls $O/pw/*/*counter_collection.csv $O/pw/*counter_collection.csv 2>/dev/null | head -1) > $O/conv_hbm_traffic_pmc.json 2> $O/pmc_traffic.err
head -c 400 $O/conv_hbm_traffic_pmc.json; echo; tail -2 $O/pmc_traffic.err
rm -rf $O/pf $O/pw
