#!/bin/bash
# round-3 GPU session 3: second-generation rotation kernel: numerics, then timing A/B (eig kernel x chains)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
echo "== pytest svd subset (eig2 default)" > $O/s3_pytest.log
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_golden.py -x -q -m gpu -k "tsvd or tsplit or svd or split" >> $O/s3_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc" >> $O/s3_pytest.log
if [ $rc -ne 0 ]; then tail -30 $O/s3_pytest.log; exit 1; fi
echo "== timing" > $O/s3_svd.log
for eig in 1 2; do for nc in 1 4; do
  echo "-- MPSK_SVD_EIG=$eig MPSK_SVD_CHAINS=$nc" >> $O/s3_svd.log
  MPSK_SVD_EIG=$eig MPSK_SVD_CHAINS=$nc timeout -k 10 300 python tools/svd_probe.py 1024,4096 graded6 uniform 2>&1 | grep -v amdgpu.ids >> $O/s3_svd.log || echo "FAILED" >> $O/s3_svd.log
done; done
cd /tmp && export TMPDIR=/tmp
for nc in 1 4; do
  rm -rf $O/st$nc
  MPSK_SVD_CHAINS=$nc timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st$nc -- python3 $R/tools/svd_once.py 4096 graded6 > $O/s3_st$nc.log 2>&1
  f=$(find $O/st$nc -name "*kernel_stats.csv" | head -1)
  echo "== chains $nc stats" >> $O/s3_stats.log
  head -8 $f | cut -c1-160 >> $O/s3_stats.log
  rm -rf $O/st$nc
done
cat $O/s3_pytest.log | tail -3; cat $O/s3_svd.log; cat $O/s3_stats.log
