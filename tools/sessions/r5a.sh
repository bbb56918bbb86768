#!/bin/bash
# round-5 GPU session A (FIRST GPU minutes of the round): the kernels written in round 4 without GPU time -- wgrad_v3l.h / wgrad_ql.h ("lean" weight
# gradients: LDS-DMA addresses once per workgroup, ReLU as a template parameter, bias gradient through v_dot2; CPU: tests/test_hipemu_cpu.py,
# bit-identical to the shipped kernels under the interpreter). 1. parity on the GPU (opt-in tests), 2. layer tables A/B, 3. step A/B.
# ~45 GPU-minutes as written: split it (sections 1-3 | 3b-3d | 4) if the budget asks for it.
# Decision rule: a switch becomes the default when its layer table is faster on the same box and the step is not slower.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5a
mkdir -p $O
( time SG_EXPERIMENTAL=1 timeout 400 python -m pytest tests/test_conv_v2_gpu.py tests/test_quad_gpu.py -q -p no:cacheprovider --maxfail=10 -k "lean" 2>&1 | tail -8 ) > $O/pytest_lean.txt 2>&1
cat $O/pytest_lean.txt | cut -c1-250
for f in 0 1 2 1p 2p; do      # wgrad_q: 1 = lean, 2 = lean + register pipeline over the k-steps (wgrad_v3 has 0 / 1 only); p = + SG_MFMA_PRIO=1 (s_setprio around the MFMA clusters)
  pr=0; case $f in *p) pr=1;; esac; l=${f%p}
  [ $l -lt 2 ] && ( SG_MFMA_PRIO=$pr SG_WGRAD_V3_LEAN=$l timeout 300 python tools/conv_bench.py --batch 256 2>&1 ) > $O/conv_bench_v3lean$f.txt 2>&1
  ( SG_MFMA_PRIO=$pr SG_WGRAD_Q_LEAN=$l timeout 300 python tools/quad_bench.py --batch 256 2>&1 ) > $O/quad_bench_qlean$f.txt 2>&1
  [ $l -lt 2 ] && tail -4 $O/conv_bench_v3lean$f.txt | cut -c1-200; tail -6 $O/quad_bench_qlean$f.txt | cut -c1-200
done
for f in 0 1; do
  ( SG_WGRAD_V3_LEAN=$f SG_WGRAD_Q_LEAN=$f timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_lean$f.json 2> $O/bench_lean$f.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/bench_lean$f.json") if l.startswith("{")][-1])
    r=j["roofline"]
    pk=r["per_kernel"]
    print("LEAN=$f", j["value"], "img/s", j["ms_per_step"], "ms; conv ms", r["conv_ms_per_step"], {k: (v["ms_per_step"], v["executed_tflops"]) for k, v in pk.items() if "wgrad" in k}, "losses", j["last_step_losses"])
except Exception as e:
    print("failed", e)
PY
  tail -2 $O/bench_lean$f.err | cut -c1-200
done
# 3b. conv_q.h with the weights three taps ahead (SG_CONV_Q_LA3=1; CPU: bit-identical to the shipped loop under the interpreter)
for f in 1 2 3; do ( SG_CONV_Q_LA3=$f timeout 300 python -m pytest tests/test_quad_gpu.py -q -p no:cacheprovider -k "conv_q and not wgrad" 2>&1 | tail -3 ) > $O/pytest_la3_$f.txt 2>&1; cat $O/pytest_la3_$f.txt | cut -c1-200; done
for f in 0 3 1 2 1p 2p; do      # 0: shipped loop, 3: one-sided patch halo (-8 % staged bytes: the LDS-DMA-ingest test), 1: weights three taps ahead, 2: taps in pairs (one barrier per pair); p = + SG_MFMA_PRIO=1
  pr=0; case $f in *p) pr=1;; esac
  ( SG_MFMA_PRIO=$pr SG_CONV_Q_LA3=${f%p} timeout 300 python tools/quad_bench.py --batch 256 2>&1 ) > $O/quad_bench_la3_$f.txt 2>&1; tail -6 $O/quad_bench_la3_$f.txt | cut -c1-200
done
# 3b'. conv_q.h with 512-pixel tiles (SG_CONV_Q_BJ=512: -35 % staged bytes per MFMA at two workgroups per CU), alone and with the one-sided halo
( SG_CONV_Q_BJ=5f timeout 300 python -m pytest tests/test_quad_gpu.py -q -p no:cacheprovider -k "conv_q and not wgrad" 2>&1 | tail -3 ) > $O/pytest_bj512.txt 2>&1; cat $O/pytest_bj512.txt | cut -c1-200
for cfg in "SG_CONV_Q_BJ=512" "SG_CONV_Q_BJ=512 SG_CONV_Q_LA3=3"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  ( env $cfg timeout 300 python tools/quad_bench.py --batch 256 2>&1 ) > $O/quad_bench_$tag.txt 2>&1; tail -6 $O/quad_bench_$tag.txt | cut -c1-200
done
# 3c. conv_v4.h with four weight buffers / three taps ahead (SG_CONV_V4_LA3=1) where three workgroups still fit a CU
( SG_CONV_V4_LA3=1 timeout 300 python -m pytest tests/test_conv_v2_gpu.py -q -p no:cacheprovider -k "conv_v4 or fused_skip" 2>&1 | tail -3 ) > $O/pytest_v4la3.txt 2>&1; cat $O/pytest_v4la3.txt | cut -c1-200
for f in 0 1 1p; do
  pr=0; case $f in *p) pr=1;; esac
  ( SG_MFMA_PRIO=$pr SG_CONV_V4_LA3=${f%p} timeout 300 python tools/conv_bench.py --batch 256 2>&1 ) > $O/conv_bench_v4la3_$f.txt 2>&1; tail -12 $O/conv_bench_v4la3_$f.txt | cut -c1-200
done
# 3d. everything that won its layer table, together: step A/B (edit the list)
for cfg in "SG_NOOP=1" "SG_WGRAD_V3_LEAN=1 SG_WGRAD_Q_LEAN=1 SG_CONV_Q_LA3=1 SG_CONV_V4_LA3=1"; do
  tag=$(echo "$cfg" | tr ' =' '__' | cut -c1-40)
  ( env $cfg timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_all_$tag.json 2> $O/bench_all_$tag.err
  echo "$cfg: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_all_$tag.json | head -1) $(grep -o '"conv_ms_per_step": [0-9.]*' $O/bench_all_$tag.json | head -1)"
done
# 4. FID leg (bf16 Inception): conv_v2 for InceptionV3's 128 / 160-cout 1x7 / 7x1 layers (SG_CONV_V2_MIN_TILES / SG_CONV_V2_PAD_TILES), and the
#    frozen-network weight-image cache (tools/fid_leg.py runs G.eval(): SG_EVAL_CACHE=1 opts in, 0 is the round-4 behaviour)
( timeout 300 python -m pytest tests/test_eval_gpu.py -q -p no:cacheprovider 2>&1 | tail -3 ) > $O/pytest_eval.txt 2>&1; cat $O/pytest_eval.txt | cut -c1-200
for cfg in "SG_EVAL_CACHE=0" "SG_EVAL_CACHE=1" "SG_EVAL_CACHE=1 SG_CONV_V2_MIN_TILES=128" "SG_EVAL_CACHE=1 SG_CONV_V2_MIN_TILES=128 SG_CONV_V2_PAD_TILES=1"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  ( env $cfg timeout 300 python tools/fid_leg.py --samples 10240 --dtype bf16 ) > $O/fid_$tag.json 2> $O/fid_$tag.err
  echo "$cfg: $(grep -o '"value": [0-9.]*' $O/fid_$tag.json | head -1)"; tail -1 $O/fid_$tag.err | cut -c1-200
done
( SG_CONV_V2_MIN_TILES=128 SG_CONV_V2_PAD_TILES=1 timeout 300 python -m pytest tests/test_eval_gpu.py -q -p no:cacheprovider 2>&1 | tail -3 ) > $O/pytest_eval_v2.txt 2>&1; cat $O/pytest_eval_v2.txt | cut -c1-200
# 5. counters that tell the vector pipe from the LDS-DMA ingest (separate passes, --kernel-trace only: MI355X_MICROARCH.md's recipe): weight-gradient layer tables,
#    shipped vs lean -- SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES falls with the lean kernels whatever happens to the time; SQ_VALU_MFMA_BUSY_CYCLES per clock is the verdict
R=$(pwd)
for f in 0 1; do
  ( cd /tmp && SG_WGRAD_V3_LEAN=$f SG_WGRAD_Q_LEAN=$f timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace -d $R/$O/pmc_lean$f -o pmc --output-format csv -- python $R/tools/quad_bench.py --batch 256 ) > $O/pmc_lean$f.log 2>&1
  python tools/pmc_summary.py $(ls $O/pmc_lean$f/*/*counter_collection.csv $O/pmc_lean$f/*counter_collection.csv 2>/dev/null | head -1) > $O/sq_counters_lean$f.txt 2> $O/sq_counters_lean$f.err
  grep -i "wgrad" $O/sq_counters_lean$f.txt | cut -c1-220 | head -8
done
