#!/bin/bash
# round-2 GPU session A: new full-width parity tests + whole GPU suite + baseline layer table + SQ counters of the conv kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2a
mkdir -p $O
rm -f gpurun_out/fullwidth_parity.txt
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=60 -p no:cacheprovider ) > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
cp gpurun_out/fullwidth_parity.txt $O/ 2>/dev/null
timeout 300 python tools/conv_bench.py --bias > $O/conv_bench.txt 2>&1
tail -3 $O/conv_bench.txt
R=$PWD
( cd /tmp && timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace -d $R/$O/pmc1 -o pmc1 --output-format csv -- python $R/tools/conv_bench.py --bias ) > $O/pmc1.log 2>&1
( cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $R/$O/pmc2 -o pmc2 --output-format csv -- python $R/tools/conv_bench.py --bias ) > $O/pmc2.log 2>&1
ls -R $O | head -40
timeout 400 python bench.py --steps 3 --warmup 2 --fid-samples 0 --no-cpu-baseline > $O/bench_quick.json 2> $O/bench_quick.err
tail -c 1500 $O/bench_quick.json
