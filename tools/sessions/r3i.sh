#!/bin/bash
# round-3 GPU session I: C4 at 256^2 as a bench extra, FID leg with every eligible layer forced onto the conv_v2 tile kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r3i
mkdir -p $O
( time timeout 300 python tools/extra_run.py bigdeep256_bs64_bf16 2 ) > $O/extra_bigdeep256.json 2> $O/extra_bigdeep256.err
tail -1 $O/extra_bigdeep256.json | cut -c1-700; tail -3 $O/extra_bigdeep256.err | cut -c1-300
( time SG_CONV_V2=force timeout 200 python tools/fid_leg.py --samples 5120 --dtype bf16 ) > $O/fid_v2force.json 2> $O/fid_v2force.err
( time timeout 200 python tools/fid_leg.py --samples 5120 --dtype bf16 ) > $O/fid_default.json 2> $O/fid_default.err
tail -1 $O/fid_v2force.json | cut -c1-120; tail -1 $O/fid_default.json | cut -c1-120
