#!/bin/bash
# round 3, session 22: potential of a warm-started gauge step (probe only) + split tests after the per-shape hint change
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s22
O=gpurun_out/s22
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_golden.py -x -q -m gpu -k "svd or split" > $O/pytest_svd.log 2>&1 || { tail -40 $O/pytest_svd.log; exit 1; }
tail -1 $O/pytest_svd.log
timeout -k 10 600 python tools/warm_gauge_probe.py 40 1024 6 2>&1 | grep -v amdgpu | tee $O/warm_probe.log
