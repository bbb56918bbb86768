#!/bin/bash
# CPU session (no GPU): every width-8 golden training step of tests/golden through the lane-accurate interpreter (tests/hipemu), plus the
# interpreter's own sensitivity test (a kernel with its s_waitcnt removed must fail). Output: profiles/rNN_hipemu_nets.txt
cd "$(dirname "$0")/../.." || exit 1
out=${1:-profiles/r04_hipemu_nets.txt}
{
  echo "# SG_EMU_NET=1 python -m pytest tests/test_hipemu_net_cpu.py tests/test_hipemu_cpu.py -q --durations=0   ($(date -u +%F), $(nproc) host cores)"
  SG_EMU_NET=1 python -m pytest tests/test_hipemu_net_cpu.py tests/test_hipemu_cpu.py -q --durations=0 -p no:cacheprovider 2>&1 | grep -v "^$" | grep -E "passed|failed|PASSED|FAILED|s call|s setup|Error|error" | grep -v "0.0[0-9]s\|0.[0-4][0-9]s call"
} > "$out"
cat "$out"
