#!/bin/bash
# round 3, session 20: workspace fix -- fresh-ctx tests, then the keep-half probe that faulted
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s20
O=gpurun_out/s20
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "fresh_ctx" > $O/pytest_fresh.log 2>&1 || { tail -30 $O/pytest_fresh.log; exit 1; }
tail -2 $O/pytest_fresh.log
for n in 2048 4096; do
  timeout -k 10 200 python tools/svd_half.py $n graded6 2 2>&1 | tail -1 | tee -a $O/half.log
  MPSK_SPLIT_MAXFRAC=0.76 timeout -k 10 200 python tools/svd_half.py $n graded6 3 2>&1 | tail -1 | sed "s/^/maxfrac 0.76 (r = 0.75 n): /" | tee -a $O/half.log
  MPSK_SPLIT_MAXFRAC=0.76 MPSK_SPLIT_OVERSAMPLE=0.25 timeout -k 10 200 python tools/svd_half.py $n graded6 3 2>&1 | tail -1 | sed "s/^/oversample 0.25 (r = 0.625 n): /" | tee -a $O/half.log
done
