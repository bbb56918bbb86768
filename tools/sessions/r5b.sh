#!/bin/bash
# round-5 GPU session B = r5a.sh sections 1-3c trimmed to what discriminates: GPU parity of every SG_EXPERIMENTAL case, then same-box layer
# tables per switch (no step A/B here: that is r5c.sh, on the winners only).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5b
mkdir -p $O
T0=$(date +%s)
( time SG_EXPERIMENTAL=1 timeout 500 python -m pytest tests/test_conv_v2_gpu.py tests/test_quad_gpu.py -q -p no:cacheprovider --maxfail=20 -k "lean" 2>&1 | tail -12 ) > $O/pytest_experimental.txt 2>&1
cat $O/pytest_experimental.txt | cut -c1-250
for f in 1 2 3; do ( SG_CONV_Q_LA3=$f timeout 300 python -m pytest tests/test_quad_gpu.py -q -p no:cacheprovider -k "conv_q and not wgrad" 2>&1 | tail -3 ) > $O/pytest_la3_$f.txt 2>&1; cat $O/pytest_la3_$f.txt | cut -c1-200; done
( SG_CONV_Q_BJ=512 timeout 300 python -m pytest tests/test_quad_gpu.py -q -p no:cacheprovider -k "conv_q and not wgrad" 2>&1 | tail -3 ) > $O/pytest_bj512.txt 2>&1; cat $O/pytest_bj512.txt | cut -c1-200
( SG_CONV_V4_LA3=1 timeout 300 python -m pytest tests/test_conv_v2_gpu.py -q -p no:cacheprovider -k "conv_v4 or fused_skip" 2>&1 | tail -3 ) > $O/pytest_v4la3.txt 2>&1; cat $O/pytest_v4la3.txt | cut -c1-200
echo "tests done at $(( $(date +%s) - T0 )) s"
for cfg in "SG_NOOP=1" "SG_CONV_Q_LA3=3" "SG_CONV_Q_BJ=512" "SG_CONV_Q_BJ=512 SG_CONV_Q_LA3=3" "SG_CONV_Q_LA3=1" "SG_CONV_Q_LA3=2" "SG_WGRAD_Q_LEAN=1" "SG_WGRAD_Q_LEAN=2" \
           "SG_CONV_Q_LA3=1 SG_WGRAD_Q_LEAN=2 SG_MFMA_PRIO=1" "SG_CONV_Q_LA3=2 SG_WGRAD_Q_LEAN=1 SG_MFMA_PRIO=1" "SG_NOOP=2"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  ( env $cfg timeout 200 python tools/quad_bench.py --batch 256 2>&1 ) > $O/quad_bench_$tag.txt 2>&1
  echo "== $cfg"; tail -11 $O/quad_bench_$tag.txt | cut -c1-200
done
echo "quad tables done at $(( $(date +%s) - T0 )) s"
for cfg in "SG_NOOP=1" "SG_WGRAD_V3_LEAN=1" "SG_CONV_V4_LA3=1" "SG_WGRAD_V3_LEAN=1 SG_CONV_V4_LA3=1 SG_MFMA_PRIO=1"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  ( env $cfg timeout 300 python tools/conv_bench.py --batch 256 2>&1 ) > $O/conv_bench_$tag.txt 2>&1
  echo "== $cfg"; tail -40 $O/conv_bench_$tag.txt | cut -c1-220
done
echo "all done at $(( $(date +%s) - T0 )) s"
