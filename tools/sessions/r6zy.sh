#!/bin/bash
# after the fourth session: bench.py's two-rank path on one device (plumbing) and the slow full-width comparisons, on the final code
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
bash tools/sessions/r6i.sh
bash tools/sessions/r6_slow.sh
