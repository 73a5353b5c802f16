#!/bin/bash
# round 3, session 29: complex128 gauge steps at the ABI (mpsk_qrpos / mpsk_lqpos under MPSK_C128)
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s29
O=gpurun_out/s29
timeout -k 10 600 python -m pytest tests/test_gpu_complex.py tests/test_abi.py -x -q -m "gpu or not gpu" -k "complex128 or abi or header or structured_split or gemm" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "qr or lq or split" > $O/pytest2.log 2>&1 || { tail -40 $O/pytest2.log; exit 1; }
tail -1 $O/pytest2.log
