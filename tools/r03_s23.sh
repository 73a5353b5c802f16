#!/bin/bash
# round 3, session 23: complex two-site split through mpsk_tsplit: tests + a timing of complex DMRG2 sweeps
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s23
O=gpurun_out/s23
timeout -k 10 900 python -m pytest tests/test_gpu_complex.py tests/test_gpu_algorithms.py -x -q -m gpu -k "complex or cplx or structured" > $O/pytest_cplx.log 2>&1 || { tail -40 $O/pytest_cplx.log; exit 1; }
tail -2 $O/pytest_cplx.log
