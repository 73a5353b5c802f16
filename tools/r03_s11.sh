#!/bin/bash
# round 3, session 11: svd mode 3 as the default: SVD / split / driver tests, probe with the ratio-based hint, oversampling scan
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s11
O=gpurun_out/s11
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_golden.py -x -q -m gpu -k "svd or split or bond_matrix" > $O/pytest_svd.log 2>&1 || { tail -40 $O/pytest_svd.log; exit 1; }
tail -2 $O/pytest_svd.log
MPSK_SVD_DEBUG=1 timeout -k 10 600 python tools/split_probe.py 3 > $O/probe.log 2> $O/probe.err || { tail -20 $O/probe.err; exit 1; }
cat $O/probe.log
for ov in 0.25 0.375 0.5 0.75 1.0; do
  MPSK_SPLIT_OVERSAMPLE=$ov timeout -k 10 120 python tools/svd_once.py 4096 graded6 3 3 2>&1 | tail -1 | sed "s/^/oversample=$ov /" | tee -a $O/oversample.log
done
timeout -k 10 900 python -m pytest tests/test_gpu_traces.py tests/test_gpu_algorithms.py -x -q -m gpu > $O/pytest_drivers.log 2>&1 || { tail -40 $O/pytest_drivers.log; exit 1; }
tail -2 $O/pytest_drivers.log
