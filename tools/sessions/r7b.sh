#!/bin/bash
# fifth session: flat tile tables in sg_sn_forward, data-gradient image transposed from the forward image
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r7b; mkdir -p $O
timeout 600 python -m pytest tests/test_sn_gpu.py tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_quad_gpu.py -q -m gpu -rf -p no:cacheprovider 2>&1 | grep -E "^FAILED|passed|failed" | tail -5
timeout 200 python tools/sn_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/sn_bench.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt --output-format csv -- python $R/tools/sn_bench.py --iters 10 ) > $O/sn_traced.txt 2>&1
python tools/kt_summary.py $(find $O/kt -name "*kernel_trace.csv" | head -1) 30 > $O/sn_kerneltrace.txt 2>&1
rm -rf $O/kt
cat $O/sn_kerneltrace.txt | cut -c1-160
