#!/bin/bash
# Round 5: microbenchmark + kernel trace of the csrc/ext augmentation kernels (no kernel of the benchmarked step involved)
mkdir -p gpurun_out/r5k
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r5k/prof -o aug -- python $GRAFT_REPO_ROOT/tools/aug_bench.py > $GRAFT_REPO_ROOT/gpurun_out/r5k/aug_bench_prof.txt 2>&1
cd $GRAFT_REPO_ROOT
timeout 100 python tools/aug_bench.py > gpurun_out/r5k/aug_bench.txt 2>&1
cat gpurun_out/r5k/aug_bench.txt
