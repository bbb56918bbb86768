#!/bin/bash
# Round 5: regression run over the GPU test files the second session's host-side changes reach but had not re-run (worker sampling path, style_ops GPU check,
# two-rank plumbing, the full-width BigGAN step through the re-structured attention block)
mkdir -p gpurun_out/r5p
run() { ( time timeout $1 python -m pytest $2 -x -q ${3:+-k "$3"} ) > gpurun_out/r5p/$4.txt 2>&1; echo "rc=$?" >> gpurun_out/r5p/$4.txt; grep -E "passed|failed|rc=" gpurun_out/r5p/$4.txt | tail -2; }
run 60 tests/test_heads_gpu.py "" heads
run 40 tests/test_style_gpu.py "" style
run 60 tests/test_fullwidth_gpu.py "step_vs_golden and biggan128w" fullwidth_biggan128w
run 60 tests/test_dist_gpu.py "" dist
