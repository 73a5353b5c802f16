#!/bin/bash
# round 3, session 28: in-step solve / Gram (trsm path) against the GEMM path of CholeskyQR3 at small block counts
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s28
O=gpurun_out/s28
for shape in "512 256" "768 256" "1024 512" "1536 512" "1024 256" "2048 1024"; do
  for t in 1 0; do
    MPSK_CQ_TRSM=$t timeout -k 10 100 python tools/qr_only.py $shape 2>&1 | grep qrpos | cut -c1-60 | sed "s/^/trsm=$t /" | tee -a $O/trsm.log
  done
done
