#!/bin/bash
# round 3, session 16: warm-started split (mpsk_ctx_split_hint): tests, config-4 sweeps
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s16
O=gpurun_out/s16
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_golden.py tests/test_gpu_traces.py -x -q -m gpu -k "svd or split or bond_matrix or trace" > $O/pytest_svd.log 2>&1 || { tail -40 $O/pytest_svd.log; exit 1; }
tail -2 $O/pytest_svd.log
C4_VERBOSE=1 timeout -k 10 600 python tools/bench_configs.py c4sweep:64:1024:3 > $O/c4_L64.log 2>&1
grep "c4sweep" $O/c4_L64.log
