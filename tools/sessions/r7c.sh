#!/bin/bash
# fifth session: 1024-thread k_sn_v / k_sn_u; is the traced C3 step GPU-bound (tools/kt_gaps.py)? same-box step time
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r7c; mkdir -p $O
timeout 600 python -m pytest tests/test_sn_gpu.py tests/test_model_gpu.py -q -m gpu -rf -p no:cacheprovider 2>&1 | grep -E "^FAILED|passed|failed" | tail -5
timeout 200 python tools/sn_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/sn_bench.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt --output-format csv -- python $R/bench.py --steps 9 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_traced.json 2> $O/bench_traced.err
KT=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python tools/kt_summary.py $KT 130 > $O/kerneltrace.txt 2>&1
python tools/kt_gaps.py $KT 6 > $O/gaps.txt 2>&1
rm -rf $O/kt
cat $O/gaps.txt | cut -c1-200
B="python bench.py --steps 10 --warmup 3 --no-extras --fid-samples 0 --no-cpu-baseline"
for z in 1 2; do
  timeout 300 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c3', d['ms_per_step'], d['roofline_hbm']['spectral_norm'])"
done
