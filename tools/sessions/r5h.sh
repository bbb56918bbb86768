#!/bin/bash
# round-5 GPU session H: single-pass flash attention forward (deferred rescale): parity, tools/attn_bench.py (base = the build of session r5d: two passes, 8x unrolled), the step.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5h
mkdir -p $O
T0=$(date +%s)
( timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py -q -p no:cacheprovider -x -k "attention or attn or biggan32" 2>&1 | tail -3 ) > $O/pytest_quick.txt 2>&1; cat $O/pytest_quick.txt | cut -c1-250
for rep in 1 2; do
  for lib in base new; do
    L=""; [ $lib = base ] && L="SG_LIBSGAMD=tools/ab_libsgamd_base.so"
    ( env $L timeout 200 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids ) > $O/attn_bench_${lib}_$rep.txt 2>&1
    echo "== attn_bench $lib $rep"; tail -3 $O/attn_bench_${lib}_$rep.txt | cut -c1-200
  done
done
for cfg in "SG_NOOP=1" "SG_NOOP=2"; do
  tag=$(echo "$cfg" | tr ' =/' '___' | cut -c1-70)
  ( env $cfg timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_$tag.json 2> $O/bench_$tag.err
  echo "$cfg: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$tag.json | head -1) $(grep -o '"conv_ms_per_step": [0-9.]*' $O/bench_$tag.json | head -1) $(grep -o '"attention_scores": {[^}]*}' $O/bench_$tag.json | head -1)"
  tail -1 $O/bench_$tag.err | cut -c1-200
done
echo "all done at $(( $(date +%s) - T0 )) s"
