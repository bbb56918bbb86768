#!/bin/bash
# round-5 GPU session F: attention kernels with their inner loops no longer unrolled eight times (one wave per SIMD -> two / three, no AGPR round trips): parity,
# tools/attn_bench.py against the previous build on one box; spectral-norm backward block count; the step; C4 at 256^2.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5f
mkdir -p $O
T0=$(date +%s)
( timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_quad_gpu.py tests/test_blocks_gpu.py -q -p no:cacheprovider -x -k "attention or attn or bn_statistics or biggan32 or resgan32" 2>&1 | tail -5 ) > $O/pytest_quick.txt 2>&1; cat $O/pytest_quick.txt | cut -c1-250
echo "quick tests done at $(( $(date +%s) - T0 )) s"
for rep in 1 2; do
  for lib in base new; do
    L=""; [ $lib = base ] && L="SG_LIBSGAMD=tools/ab_libsgamd_base.so"
    ( env $L timeout 200 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids ) > $O/attn_bench_${lib}_$rep.txt 2>&1
    echo "== attn_bench $lib $rep"; cat $O/attn_bench_${lib}_$rep.txt | cut -c1-200
  done
done
echo "attention done at $(( $(date +%s) - T0 )) s"
for nb in 512 1024 2048 4096; do
  ( SG_SNB_BLOCKS=$nb timeout 200 python tools/sn_bench.py 2>&1 | grep -v amdgpu.ids | tail -2 ) > $O/sn_bench_nb$nb.txt 2>&1; echo "SG_SNB_BLOCKS=$nb"; cat $O/sn_bench_nb$nb.txt | cut -c1-250
done
for cfg in "SG_NOOP=1" "SG_NOOP=2"; do
  tag=$(echo "$cfg" | tr ' =/' '___' | cut -c1-70)
  ( env $cfg timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_$tag.json 2> $O/bench_$tag.err
  echo "$cfg: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$tag.json | head -1) $(grep -o '"conv_ms_per_step": [0-9.]*' $O/bench_$tag.json | head -1) $(grep -o '"attention_scores": {[^}]*}' $O/bench_$tag.json | head -1) $(grep -o '"spectral_norm": {[^}]*}' $O/bench_$tag.json | head -1)"
  tail -1 $O/bench_$tag.err | cut -c1-200
done
( timeout 300 python tools/extra_run.py bigdeep256_bs64_bf16 3 ) > $O/extra_bigdeep256.json 2> $O/extra_bigdeep256.err
echo "bigdeep256: $(grep -o '"images_per_sec": [0-9.]*' $O/extra_bigdeep256.json) $(grep -o '"ms_per_step": [0-9.]*' $O/extra_bigdeep256.json) $(grep -o '"conv_ms_per_step": [0-9.]*' $O/extra_bigdeep256.json)"
echo "all done at $(( $(date +%s) - T0 )) s"
