#!/bin/bash
# round-5 GPU session D: the "lean" forward loop of conv_q.h (per-view address tables, scalar DMA offsets) against the previous build on one box; spectral norm in
# Infinity-Cache-sized runs; the step on the new defaults; the whole GPU suite with durations.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5d
mkdir -p $O
T0=$(date +%s)
( timeout 600 python -m pytest tests/test_quad_gpu.py tests/test_kernels_gpu.py -q -p no:cacheprovider -x 2>&1 | tail -5 ) > $O/pytest_quick.txt 2>&1; cat $O/pytest_quick.txt | cut -c1-250
echo "quick tests done at $(( $(date +%s) - T0 )) s"
for rep in 1 2; do
  for lib in base new; do
    L=""; [ $lib = base ] && L="SG_LIBSGAMD=tools/ab_libsgamd_base.so"
    ( env $L timeout 200 python tools/quad_bench.py --batch 256 2>&1 ) > $O/quad_bench_${lib}_$rep.txt 2>&1
    echo "== quad_bench $lib $rep: $(grep '^sum' $O/quad_bench_${lib}_$rep.txt | cut -c1-200)"
  done
done
echo "quad tables done at $(( $(date +%s) - T0 )) s"
for mb in 0 48 96 160; do
  ( SG_SN_CHUNK_MB=$mb timeout 200 python tools/sn_bench.py 2>&1 | grep -v amdgpu.ids ) > $O/sn_bench_$mb.txt 2>&1; cat $O/sn_bench_$mb.txt | cut -c1-250
done
echo "sn done at $(( $(date +%s) - T0 )) s"
for cfg in "SG_NOOP=1" "SG_LIBSGAMD=tools/ab_libsgamd_base.so" "SG_SN_CHUNK_MB=0" "SG_CONV_V2_MIN_TILES=128" "SG_NOOP=2"; do
  tag=$(echo "$cfg" | tr ' =/' '___' | cut -c1-70)
  ( env $cfg timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_$tag.json 2> $O/bench_$tag.err
  echo "$cfg: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$tag.json | head -1) $(grep -o '"conv_ms_per_step": [0-9.]*' $O/bench_$tag.json | head -1) $(grep -o '"spectral_norm": {[^}]*}' $O/bench_$tag.json | head -1)"
  tail -1 $O/bench_$tag.err | cut -c1-200
done
echo "step A/B done at $(( $(date +%s) - T0 )) s"
( time timeout 1300 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=70 2>&1 | tail -110 ) > $O/pytest_gpu_full.txt 2>&1
tail -8 $O/pytest_gpu_full.txt | cut -c1-200
echo "all done at $(( $(date +%s) - T0 )) s"
