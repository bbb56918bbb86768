#!/bin/bash
# round-3 GPU session J (closing): whole GPU suite, smoke(), default bench, PMC traffic passes, kernel traces -- on the final code state
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r3j
mkdir -p $O
rm -f gpurun_out/fullwidth_parity.txt
( time timeout 1800 python -m pytest tests -m gpu -q --maxfail=60 -p no:cacheprovider --durations=8 ) > $O/pytest_gpu.txt 2>&1
grep -E "^FAILED|passed|failed|error" $O/pytest_gpu.txt | cut -c1-200 | tail -20
cp gpurun_out/fullwidth_parity.txt gpurun_out/fid_backend_2048.txt $O/ 2>/dev/null
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.txt 2>&1
tail -3 $O/smoke.txt
( time timeout 900 python bench.py --strict ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 1200 $O/bench_default.json; echo; tail -3 $O/bench_default.err | cut -c1-300
B2="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --fid-samples 0"
( cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/pf -o pf --output-format csv -- $B2 ) > $O/pf.log 2>&1
( cd /tmp && timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$O/pw -o pw --output-format csv -- $B2 ) > $O/pw.log 2>&1
python tools/pmc_traffic.py $(ls $O/pf/*/*counter_collection.csv $O/pf/*counter_collection.csv 2>/dev/null | head -1) $(ls $O/pw/*/*counter_collection.csv $O/pw/*counter_collection.csv 2>/dev/null | head -1) > $O/conv_hbm_traffic_pmc.json 2> $O/pmc_traffic.err
head -c 600 $O/conv_hbm_traffic_pmc.json; echo; tail -2 $O/pmc_traffic.err
rm -rf $O/pf $O/pw
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_traced.json 2> $O/bench_traced.err
python tools/kt_summary.py $(ls $O/kt/*/*kernel_trace.csv $O/kt/*kernel_trace.csv 2>/dev/null | head -1) 120 > $O/kerneltrace.txt 2>&1
rm -rf $O/kt
for W in bigdeep256_bs64_bf16 wgangp128_bs64_bf16 bigdeep128_bs256_bf16; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kx -o kx --output-format csv -- python $R/tools/extra_run.py $W 2 ) > $O/extra_$W.json 2> $O/extra_$W.err
  python tools/kt_summary.py $(ls $O/kx/*/*kernel_trace.csv $O/kx/*kernel_trace.csv 2>/dev/null | head -1) 40 > $O/kerneltrace_extra_$W.txt 2>&1
  rm -rf $O/kx
done
head -8 $O/kerneltrace_extra_bigdeep256_bs64_bf16.txt | cut -c1-150
