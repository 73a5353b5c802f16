#!/bin/bash
# round 3, session 9: intra/cross step order in jacobi_eig2_kernel, skipping intra steps
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s9
O=gpurun_out/s9
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_golden.py -x -q -m gpu -k "svd or split" > $O/pytest_svd.log 2>&1 || { tail -30 $O/pytest_svd.log; exit 1; }
tail -2 $O/pytest_svd.log
for r in 0 1e-3 1e-2 1e-1 1; do
  for n in 4096 2048; do
    MPSK_SVD_INTRA=$r timeout -k 10 120 python tools/svd_once.py $n graded6 3 2>&1 | tail -1 | sed "s/^/intra=$r /" | tee -a $O/intra.log
  done
  MPSK_SVD_INTRA=$r timeout -k 10 120 python tools/svd_once.py 4096 uniform 3 2>&1 | tail -1 | sed "s/^/intra=$r /" | tee -a $O/intra.log
done
