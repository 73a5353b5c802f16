#!/bin/bash
# round-3 GPU session 5: reworked diagonal factorization of cq_step_kernel: numerics, then timing
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_dist.py -x -q -m gpu -k "qr or lq or chol or gauge or defer or dist or rccl or sharded" > $O/s5_pytest.log 2>&1
rc=$?; tail -3 $O/s5_pytest.log
if [ $rc -ne 0 ]; then grep -n "assert\|Error" $O/s5_pytest.log | head; exit 1; fi
(timeout -k 10 100 python tools/qr_only.py 2048 1024; timeout -k 10 100 python tools/qr_only.py 768 256; timeout -k 10 100 python tools/qr_only.py 1024 512; timeout -k 10 100 python tools/qr_only.py 4096 4096 3) 2>&1 | grep -v amdgpu > $O/s5_qr.log
cat $O/s5_qr.log
timeout -k 10 300 python tools/bench_configs.py c2 c3 2>&1 | grep -v amdgpu | tail -3 > $O/s5_cfg.log; cat $O/s5_cfg.log
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-tolerance-sweep --early-sweeps 0 2>/dev/null | cut -c1-200 > $O/s5_bench.log; cat $O/s5_bench.log
