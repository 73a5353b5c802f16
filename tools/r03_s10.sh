#!/bin/bash
# round 3, session 10: truncation-aware mpsk_tsplit (svd mode 3)
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s10
O=gpurun_out/s10
MPSK_SVD_DEBUG=1 timeout -k 10 600 python tools/split_probe.py 2 > $O/probe.log 2> $O/probe.err || { tail -20 $O/probe.err; tail -5 $O/probe.log; exit 1; }
cat $O/probe.log
grep "subspace stage" $O/probe.err | tail -30
