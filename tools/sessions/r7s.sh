#!/bin/bash
# fifth session, closing call: PMC traffic passes keyed to the final csrc hash (a comment in norm.hip changed after r7zz), batch-norm kernel tests, bench with the driver's flags
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r7s; mkdir -p $O
B2="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --fid-samples 0"
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/pf -o pf --output-format csv -- $B2 ) > $O/pf.log 2>&1
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/$O/pw -o pw --output-format csv -- $B2 ) > $O/pw.log 2>&1
python tools/pmc_traffic.py $(find $O/pf -name "*counter_collection.csv" | head -1) $(find $O/pw -name "*counter_collection.csv" | head -1) > $O/conv_hbm_traffic_pmc.json 2> $O/pmc_traffic.err
head -c 260 $O/conv_hbm_traffic_pmc.json; echo; tail -2 $O/pmc_traffic.err
[ -s $O/conv_hbm_traffic_pmc.json ] && cp $O/conv_hbm_traffic_pmc.json profiles/r06_conv_hbm_traffic_pmc.json
rm -rf $O/pf $O/pw
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tail -2
( timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --fid-samples 0 --no-cpu-baseline 2>/dev/null | tail -1 ) > $O/bench_20steps.json
python -c "import json; d=json.loads(open('$O/bench_20steps.json').read()); r=d['roofline']; print('20 steps / 5 warm-up:', d['ms_per_step'], d['value'], 'pmc_stale', r['pmc_stale'], r['csrc_sha16'], 'traffic', r['traffic'], 'dominant', r['dominant_kernel']['pmc_bytes_over_algorithmic'])"
