#!/bin/bash
# round-2 GPU session R: SQ counters of the current contraction kernels (conv_v4, conv_v3, wgrad_v3, ...) over the layer table
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2r
mkdir -p $O
R=$PWD
( cd /tmp && timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace -d $R/$O/pmc1 -o pmc1 --output-format csv -- python $R/tools/conv_bench.py --bias ) > $O/pmc1.log 2>&1
( cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $R/$O/pmc2 -o pmc2 --output-format csv -- python $R/tools/conv_bench.py --bias ) > $O/pmc2.log 2>&1
python tools/pmc_summary.py $O/pmc1/pmc1_counter_collection.csv $O/pmc2/pmc2_counter_collection.csv > $O/conv_sq_counters.txt 2> $O/pmc_summary.err
cut -c1-100,330-460 $O/conv_sq_counters.txt | head -40
rm -f $O/pmc1/*.csv $O/pmc2/*.csv
