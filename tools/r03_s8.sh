#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_golden.py -x -q -m gpu -k "tsvd or tsplit or svd or split" > $O/s8_pytest.log 2>&1; tail -2 $O/s8_pytest.log
MPSK_SVD_STAMPS=1 MPSK_SVD_DEBUG=1 MPSK_SVD_CHAINS=1 timeout -k 10 120 python tools/svd_once.py 4096 graded6 2>&1 | grep -E "stamps|tsplit" | head -3
timeout -k 10 200 python tools/svd_probe.py 1024,4096 graded6 uniform 2>&1 | grep tsplit
