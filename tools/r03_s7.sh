#!/bin/bash
# round-3 GPU session 7: which side paces a phase of jacobi_eig2_kernel?  (diagnostic builds: results are wrong on purpose)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
: > $O/s7.log
for lib in "" $R/mpskit.jl_amd/ab/libmpsk_eigd1.so $R/mpskit.jl_amd/ab/libmpsk_eigd2.so; do
  echo "== MPSK_LIB=$lib" >> $O/s7.log
  MPSK_LIB=$lib MPSK_SVD_STAMPS=1 MPSK_SVD_DEBUG=1 MPSK_SVD_CHAINS=1 timeout -k 10 120 python tools/svd_once.py 4096 graded6 2>&1 | grep -E "stamps" | head -2 >> $O/s7.log
done
cat $O/s7.log
