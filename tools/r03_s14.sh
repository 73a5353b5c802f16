#!/bin/bash
# round 3, session 14: config-4 sweeps at L = 64 with the truncation-aware split (per-call log), split tests, probe
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s14
O=gpurun_out/s14
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_golden.py -x -q -m gpu -k "svd or split or bond_matrix" > $O/pytest_svd.log 2>&1 || { tail -40 $O/pytest_svd.log; exit 1; }
tail -2 $O/pytest_svd.log
C4_VERBOSE=1 timeout -k 10 600 python tools/bench_configs.py c4sweep:64:1024:3 > $O/c4_L64.log 2>&1
grep "c4sweep" $O/c4_L64.log
timeout -k 10 600 python tools/split_probe.py 3 > $O/probe.log 2>&1
cat $O/probe.log
