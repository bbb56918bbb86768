#!/bin/bash
# round-5 GPU session I: the small data-gradient GEMMs of the conditional batch norms with their contraction sliced over the batches of one launch (functional.gemm_dgrad_rows):
# parity of every network that has them, the micro-benchmark, step A/B, then bench.py --strict on the final code.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5i
mkdir -p $O
T0=$(date +%s)
( timeout 600 python -m pytest tests/test_blocks_gpu.py tests/test_model_gpu.py tests/test_kernels_gpu.py -q -p no:cacheprovider -x -k "biggan32 or resgan32 or sngan32 or bigdeep32 or linear or head" 2>&1 | tail -3 ) > $O/pytest_quick.txt 2>&1; cat $O/pytest_quick.txt | cut -c1-250
( timeout 200 python tools/cbn_gemm_bench.py 2>&1 | grep -v amdgpu.ids ) > $O/cbn_gemm_bench.txt 2>&1; cat $O/cbn_gemm_bench.txt | cut -c1-200
for cfg in "SG_DGRAD_SPLITK=0" "SG_DGRAD_SPLITK=1" "SG_DGRAD_SPLITK=0 SG_NOOP=2" "SG_DGRAD_SPLITK=1 SG_NOOP=2"; do
  tag=$(echo "$cfg" | tr ' =/' '___' | cut -c1-70)
  ( env $cfg timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_$tag.json 2> $O/bench_$tag.err
  echo "$cfg: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$tag.json | head -1) $(grep -o '"conv_ms_per_step": [0-9.]*' $O/bench_$tag.json | head -1) $(grep -o '"gemm_ms_per_step": [0-9.]*' $O/bench_$tag.json | head -1)"
done
echo "A/B done at $(( $(date +%s) - T0 )) s"
( time timeout 900 python bench.py --strict ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.json; echo; tail -3 $O/bench_default.err | cut -c1-200
echo "all done at $(( $(date +%s) - T0 )) s"
