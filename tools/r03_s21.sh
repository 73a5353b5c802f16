#!/bin/bash
# round 3, session 21: adaptive oversampling (keep-half splits take the stage with r = 5n/8): full GPU suite + probes
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s21
O=gpurun_out/s21
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
for n in 1024 2048 4096; do
  timeout -k 10 200 python tools/svd_half.py $n graded6 2 2>&1 | tail -1 | tee -a $O/half.log
  timeout -k 10 200 python tools/svd_half.py $n graded6 3 2>&1 | tail -1 | tee -a $O/half.log
done
