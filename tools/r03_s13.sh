#!/bin/bash
# round 3, session 13: per-call diagnostics of the config-4 splits
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s13
O=gpurun_out/s13
C4_VERBOSE=1 MPSK_SVD_DEBUG=1 timeout -k 10 600 python tools/bench_configs.py c4sweep:32:1024:3 > $O/c4_verbose.log 2> $O/c4_verbose.err
grep -c "split" $O/c4_verbose.log
grep "c4sweep" $O/c4_verbose.log
