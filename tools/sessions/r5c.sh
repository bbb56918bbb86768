#!/bin/bash
# round-5 GPU session C: step A/B of the layer-table winners of r5b (lean weight gradients, s_setprio, one-sided halo), FID-leg dispatch switches,
# the 16384-query attention case, C4@256 before / after the streaming attention, SQ counters of the quad layer table.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5c
mkdir -p $O
T0=$(date +%s)
( timeout 300 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "attention" 2>&1 | tail -5 ) > $O/pytest_attention.txt 2>&1; cat $O/pytest_attention.txt | cut -c1-250
echo "tests done at $(( $(date +%s) - T0 )) s"
for cfg in "SG_NOOP=1" "SG_WGRAD_V3_LEAN=1 SG_WGRAD_Q_LEAN=2" "SG_WGRAD_V3_LEAN=1 SG_WGRAD_Q_LEAN=2 SG_MFMA_PRIO=1" "SG_WGRAD_V3_LEAN=1 SG_WGRAD_Q_LEAN=2 SG_MFMA_PRIO=1 SG_CONV_Q_LA3=3" "SG_WGRAD_V3_LEAN=1 SG_WGRAD_Q_LEAN=1" "SG_NOOP=2"; do
  tag=$(echo "$cfg" | tr ' =' '__' | cut -c1-70)
  ( env $cfg timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --fid-samples 0 ) > $O/bench_$tag.json 2> $O/bench_$tag.err
  echo "$cfg: $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$tag.json | head -1) $(grep -o '"conv_ms_per_step": [0-9.]*' $O/bench_$tag.json | head -1)"
done
echo "step A/B done at $(( $(date +%s) - T0 )) s"
( timeout 300 python tools/extra_run.py bigdeep256_bs64_bf16 3 ) > $O/extra_bigdeep256.json 2> $O/extra_bigdeep256.err; tail -c 1500 $O/extra_bigdeep256.json; tail -2 $O/extra_bigdeep256.err
echo "extra done at $(( $(date +%s) - T0 )) s"
for cfg in "SG_EVAL_CACHE=0" "SG_EVAL_CACHE=1" "SG_EVAL_CACHE=1 SG_CONV_V2_MIN_TILES=128" "SG_EVAL_CACHE=1 SG_CONV_V2_MIN_TILES=128 SG_CONV_V2_PAD_TILES=1"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  ( env $cfg timeout 300 python tools/fid_leg.py --samples 10240 --dtype bf16 ) > $O/fid_$tag.json 2> $O/fid_$tag.err
  echo "$cfg: $(grep -o '"value": [0-9.]*' $O/fid_$tag.json | head -1)"; tail -1 $O/fid_$tag.err | cut -c1-200
done
( SG_CONV_V2_MIN_TILES=128 SG_CONV_V2_PAD_TILES=1 timeout 300 python -m pytest tests/test_eval_gpu.py -q -p no:cacheprovider 2>&1 | tail -3 ) > $O/pytest_eval_v2.txt 2>&1; cat $O/pytest_eval_v2.txt | cut -c1-200
echo "fid done at $(( $(date +%s) - T0 )) s"
R=$(pwd)
( cd /tmp && SG_WGRAD_Q_LEAN=2 timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace -d $R/$O/pmc_q1 -o pmc --output-format csv -- python $R/tools/quad_bench.py --batch 256 ) > $O/pmc_q1.log 2>&1
( cd /tmp && SG_WGRAD_Q_LEAN=2 timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVES --kernel-trace -d $R/$O/pmc_q2 -o pmc --output-format csv -- python $R/tools/quad_bench.py --batch 256 ) > $O/pmc_q2.log 2>&1
python tools/pmc_summary.py $(find $O/pmc_q1 $O/pmc_q2 -name "*counter_collection.csv") > $O/sq_counters_quad.txt 2> $O/sq_counters_quad.err
head -12 $O/sq_counters_quad.txt | cut -c1-400; tail -3 $O/pmc_q2.log
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -delete
echo "all done at $(( $(date +%s) - T0 )) s"
