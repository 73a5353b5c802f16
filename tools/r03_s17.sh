#!/bin/bash
# round 3, session 17: truncation-aware split when HALF of the values are kept (d = 2 chains)
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s17
O=gpurun_out/s17
for n in 2048 4096; do
  timeout -k 10 200 python tools/svd_half.py $n graded6 2 2>&1 | tail -1 | tee -a $O/half.log
  MPSK_SPLIT_MAXFRAC=0.76 timeout -k 10 200 python tools/svd_half.py $n graded6 3 2>&1 | tail -1 | sed "s/^/maxfrac 0.76 (r = 0.75 n): /" | tee -a $O/half.log
  MPSK_SPLIT_MAXFRAC=0.76 MPSK_SPLIT_OVERSAMPLE=0.25 timeout -k 10 200 python tools/svd_half.py $n graded6 3 2>&1 | tail -1 | sed "s/^/oversample 0.25 (r = 0.625 n): /" | tee -a $O/half.log
  MPSK_SPLIT_MAXFRAC=0.76 MPSK_SPLIT_OVERSAMPLE=0.375 timeout -k 10 200 python tools/svd_half.py $n graded6 3 2>&1 | tail -1 | sed "s/^/oversample 0.375 (r = 0.6875 n): /" | tee -a $O/half.log
done
