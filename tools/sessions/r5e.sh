#!/bin/bash
# round-5 GPU session E: conv_v4.h with the per-lane DMA offsets hoisted (layer table against the previous build, same box), spectral norm with the layer table
# split by kind, GradLink in the BigGAN-deep blocks (A/B through SG_GRAD_LINK), the step, the FID leg's kernel trace.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5e
mkdir -p $O
T0=$(date +%s)
( timeout 900 python -m pytest tests/test_conv_v2_gpu.py tests/test_model_gpu.py tests/test_blocks_gpu.py tests/test_fullwidth_gpu.py -q -p no:cacheprovider -x -k "conv_v4 or fused_skip or bigdeep" 2>&1 | tail -5 ) > $O/pytest_quick.txt 2>&1; cat $O/pytest_quick.txt | cut -c1-250
echo "quick tests done at $(( $(date +%s) - T0 )) s"
for rep in 1 2; do
  for lib in base new; do
    L=""; [ $lib = base ] && L="SG_LIBSGAMD=tools/ab_libsgamd_base.so"
    ( env $L timeout 300 python tools/conv_bench.py --batch 256 2>&1 ) > $O/conv_bench_${lib}_$rep.txt 2>&1
    echo "== conv_bench $lib $rep: $(grep '^sum' $O/conv_bench_${lib}_$rep.txt | cut -c1-200)"
  done
done
echo "layer tables done at $(( $(date +%s) - T0 )) s"
( timeout 200 python tools/sn_bench.py 2>&1 | grep -v amdgpu.ids ) > $O/sn_bench.txt 2>&1; cat $O/sn_bench.txt | cut -c1-250
for cfg in "SG_NOOP=1" "SG_NOOP=2"; do
  tag=$(echo "$cfg" | tr ' =/' '___' | cut -c1-70)
  ( env $cfg timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_$tag.json 2> $O/bench_$tag.err
  echo "$cfg: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$tag.json | head -1) $(grep -o '"conv_ms_per_step": [0-9.]*' $O/bench_$tag.json | head -1) $(grep -o '"spectral_norm": {[^}]*}' $O/bench_$tag.json | head -1)"
  tail -1 $O/bench_$tag.err | cut -c1-200
done
echo "step done at $(( $(date +%s) - T0 )) s"
for name in bigdeep128_bs256_bf16 bigdeep256_bs64_bf16 wgangp128_bs64_bf16; do
  for gl in 1 0; do
    ( SG_GRAD_LINK=$gl timeout 300 python tools/extra_run.py $name 3 ) > $O/extra_${name}_gl$gl.json 2> $O/extra_${name}_gl$gl.err
    echo "$name SG_GRAD_LINK=$gl: $(grep -o '"images_per_sec": [0-9.]*' $O/extra_${name}_gl$gl.json) $(grep -o '"ms_per_step": [0-9.]*' $O/extra_${name}_gl$gl.json) $(grep -o '"conv_ms_per_step": [0-9.]*' $O/extra_${name}_gl$gl.json) $(grep -o '"conv_engine_frac_of_peak": [0-9.]*' $O/extra_${name}_gl$gl.json)"
  done
done
echo "extras done at $(( $(date +%s) - T0 )) s"
R=$(pwd)
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/fid_kt -o kt --output-format csv -- python $R/tools/fid_leg.py --samples 5120 --dtype bf16 ) > $O/fid_kt.log 2>&1
python tools/kt_summary.py $(find $O/fid_kt -name "*kernel_trace.csv" | head -1) > $O/fid_leg_kerneltrace.txt 2> $O/fid_kt.err; head -25 $O/fid_leg_kerneltrace.txt | cut -c1-180
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
echo "all done at $(( $(date +%s) - T0 )) s"
