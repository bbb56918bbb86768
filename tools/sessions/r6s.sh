#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$(pwd); O=gpurun_out/r6s; mkdir -p $O
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/pf -o pf --output-format csv -- python $R/tools/conv_bench.py ) > $O/pf.log 2>&1
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$O/pw -o pw --output-format csv -- python $R/tools/conv_bench.py ) > $O/pw.log 2>&1
python tools/pmc_dispatches.py $(find $O/pf -name "*counter_collection.csv" | head -1) $(find $O/pw -name "*counter_collection.csv" | head -1) "sg_conv|sg_wgrad|k_quad|k_splitk|sg_gemm" > $O/conv_dispatch_traffic.txt 2>&1
cat $O/conv_dispatch_traffic.txt | cut -c1-150
rm -rf $O/pf $O/pw
