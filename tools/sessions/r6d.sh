#!/bin/bash
# round 6: kernel-level spectral norm against torch.nn.utils.spectral_norm; C3's D and G teacher-forced against the reference graph's rounding (quad_emu=False)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6d; mkdir -p $O
rm -f gpurun_out/reference_rounding_table.txt gpurun_out/fullwidth_parity.txt
( time timeout 900 python -m pytest tests/test_sn_gpu.py -q -m gpu -p no:cacheprovider -s 2>&1 ) > $O/pytest_sn.txt 2>&1
grep -E " passed| failed|^FAILED|^ERROR" $O/pytest_sn.txt | head
( time timeout 1200 python -m pytest tests/test_fullwidth_gpu.py -q -m gpu -p no:cacheprovider -k reference_graph_rounding 2>&1 ) > $O/pytest_refround.txt 2>&1
grep -E " passed| failed|^FAILED|^ERROR|real" $O/pytest_refround.txt | head
cp gpurun_out/reference_rounding_table.txt gpurun_out/fullwidth_parity.txt $O/ 2>/dev/null
