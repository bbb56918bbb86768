#!/bin/bash
# round-6 closing GPU session on the final code of the FIFTH session (flat spectral-norm tables, dominant-kernel-only event pairs in the timed region, batch-norm apply policy): PMC traffic passes,
# bench.py --strict, kernel trace + D-forward timeline of the step, the whole GPU suite with durations, smoke, kernel traces of the extra workloads and of the FID leg,
# the two-rank one-device plumbing run, SQ counters of the attention kernels.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r7zz
mkdir -p $O
T0=$(date +%s)
B2="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --fid-samples 0"
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/pf -o pf --output-format csv -- $B2 ) > $O/pf.log 2>&1
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$O/pw -o pw --output-format csv -- $B2 ) > $O/pw.log 2>&1
python tools/pmc_traffic.py $(find $O/pf -name "*counter_collection.csv" | head -1) $(find $O/pw -name "*counter_collection.csv" | head -1) > $O/conv_hbm_traffic_pmc.json 2> $O/pmc_traffic.err
head -c 300 $O/conv_hbm_traffic_pmc.json; echo; tail -2 $O/pmc_traffic.err
[ -s $O/conv_hbm_traffic_pmc.json ] && cp $O/conv_hbm_traffic_pmc.json profiles/r06_conv_hbm_traffic_pmc.json
rm -rf $O/pf $O/pw
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/qf -o qf --output-format csv -- python $R/tools/quad_bench.py ) > $O/qf.log 2>&1
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$O/qw -o qw --output-format csv -- python $R/tools/quad_bench.py ) > $O/qw.log 2>&1
python tools/pmc_dispatches.py $(find $O/qf -name "*counter_collection.csv" | head -1) $(find $O/qw -name "*counter_collection.csv" | head -1) "sg_conv_q" > $O/quad_dispatch_traffic_after.txt 2>&1
rm -rf $O/qf $O/qw
timeout 200 python tools/quad_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-220 > $O/quad_bench.txt
timeout 200 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 > $O/conv_layer_table.txt
timeout 200 python tools/sn_bench.py 2>&1 | grep -v amdgpu.ids > $O/sn_bench.txt
timeout 200 python tools/bn_bench.py --variants ,02 2>&1 | grep -v amdgpu.ids > $O/bn_bench.txt
echo "pmc done at $(( $(date +%s) - T0 )) s"
( time timeout 900 python bench.py --strict ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 500 $O/bench_default.json; echo; tail -4 $O/bench_default.err | cut -c1-300
echo "default bench done at $(( $(date +%s) - T0 )) s"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt --output-format csv -- python $R/bench.py --steps 9 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_traced.json 2> $O/bench_traced.err
KT=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python tools/kt_summary.py $KT 130 > $O/kerneltrace.txt 2>&1
python tools/dfwd_timeline.py $KT > $O/dfwd_timeline.txt 2>&1
python tools/kt_gaps.py $KT 6 3 2 > $O/gaps.txt 2>&1
rm -rf $O/kt
head -4 $O/kerneltrace.txt | cut -c1-150; tail -3 $O/dfwd_timeline.txt
echo "trace done at $(( $(date +%s) - T0 )) s"
( time timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=40 2>&1 | tail -75 ) > $O/pytest_gpu_full.txt 2>&1
grep -E " passed| failed|FAILED|ERROR" $O/pytest_gpu_full.txt | head -8
echo "suite done at $(( $(date +%s) - T0 )) s"
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -4 ) > $O/smoke.txt 2>&1; cat $O/smoke.txt
for name in wgangp128_bs64_bf16 bigdeep128_bs256_bf16 bigdeep256_bs64_bf16; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kx -o kx --output-format csv -- python $R/tools/extra_run.py $name 2 ) > $O/extra_$name.json 2> $O/extra_$name.err
  python tools/kt_summary.py $(find $O/kx -name "*kernel_trace.csv" | head -1) 60 > $O/kerneltrace_extra_$name.txt 2>&1
  rm -rf $O/kx
  echo "$name: $(grep -o '"images_per_sec": [0-9.]*' $O/extra_$name.json) $(head -1 $O/kerneltrace_extra_$name.txt)"
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kf -o kf --output-format csv -- python $R/tools/fid_leg.py --samples 2816 --batch 128 --dtype bf16 ) > $O/fid_leg.json 2> $O/fid_leg.err
python tools/kt_summary.py $(find $O/kf -name "*kernel_trace.csv" | head -1) 60 > $O/fid_leg_kerneltrace.txt 2>&1
rm -rf $O/kf
echo "extras + fid traces done at $(( $(date +%s) - T0 )) s"
( cd /tmp && timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace -d $R/$O/ps -o ps --output-format csv -- $B2 ) > $O/ps.log 2>&1
python tools/pmc_summary.py $(find $O/ps -name "*counter_collection.csv" | head -1) 2> $O/ps.err | grep -E "^#|kernel|sg_conv_q|sg_wgrad|sg_conv_v4|sg_conv_v3|sg_conv_sk" > $O/step_sq_counters.txt; cut -c1-70,100-400 $O/step_sq_counters.txt | head -30
rm -rf $O/ps
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace -d $R/$O/pa -o pa --output-format csv -- python $R/tools/attn_bench.py --iters 3 ) > $O/pa.log 2>&1
python tools/pmc_summary.py $(find $O/pa -name "*counter_collection.csv" | head -1) 2> $O/pa.err | grep -E "^#|kernel|k_attn|k_maxpool" > $O/attention_sq_counters.txt; cut -c1-60,90-400 $O/attention_sq_counters.txt | head -12
rm -rf $O/pa
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
echo "all done at $(( $(date +%s) - T0 )) s"
