#!/bin/bash
# round-5 GPU session G: kernel trace of the step on the current code (what is left outside the convolution engine), attention inner loops unrolled twice vs once,
# the statistics-offer test again.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5g
mkdir -p $O
T0=$(date +%s)
( timeout 600 python -m pytest tests/test_quad_gpu.py tests/test_conv_v2_gpu.py -q -p no:cacheprovider -x -k "bn_statistics or fused_skip" 2>&1 | tail -3 ) > $O/pytest_quick.txt 2>&1; cat $O/pytest_quick.txt | cut -c1-250
for rep in 1 2; do
  for lib in u1 u2; do
    L=""; [ $lib = u2 ] && L="SG_LIBSGAMD=tools/ab_libsgamd_attn_u2.so"
    ( env $L timeout 200 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids ) > $O/attn_bench_${lib}_$rep.txt 2>&1
    echo "== attn_bench unroll $lib $rep"; tail -3 $O/attn_bench_${lib}_$rep.txt | cut -c1-200
  done
done
R=$(pwd)
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/kt.log 2>&1
python tools/kt_summary.py $(find $O/kt -name "*kernel_trace.csv" | head -1) > $O/bench_kerneltrace.txt 2> $O/kt.err; head -80 $O/bench_kerneltrace.txt | cut -c1-170
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
echo "all done at $(( $(date +%s) - T0 )) s"
