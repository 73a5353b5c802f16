#!/bin/bash
# round-3 GPU session 1: new parity fixtures, chained Jacobi schedule A/B, bench with the new legs
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/s1_dev.log 2>&1
echo "== svd chains A/B" > $O/s1_svd.log
for nc in 1 2 4; do
  echo "-- MPSK_SVD_CHAINS=$nc" >> $O/s1_svd.log
  MPSK_SVD_CHAINS=$nc timeout -k 10 300 python tools/svd_probe.py 2048,4096 graded6 uniform >> $O/s1_svd.log 2>&1 || echo "FAILED nc=$nc rc=$?" >> $O/s1_svd.log
done
echo "== pytest new" > $O/s1_pytest.log
timeout -k 10 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_traces.py "tests/test_gpu_ops.py" -x -q -m gpu >> $O/s1_pytest.log 2>&1
echo "pytest rc=$?" >> $O/s1_pytest.log
echo "== bench" > $O/s1_bench.log
( time timeout -k 10 900 python bench.py --steps 2 --warmup 1 ) >> $O/s1_bench.log 2>&1
echo "bench rc=$?" >> $O/s1_bench.log
tail -5 $O/s1_svd.log; tail -5 $O/s1_pytest.log; tail -3 $O/s1_bench.log
