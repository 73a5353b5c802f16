#!/bin/bash
# round-3 GPU session 6: A/B of the cq_step diagonal-factorization changes (same box): GOLDSCHMIDT / SKIP_TILES / STAGE_STORES
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
: > $O/s6_ab.log
for rep in 1 2; do
for v in 000 100 110 101 111 011; do
  for shape in "2048 1024 30" "768 256 30"; do
    MPSK_LIB=$R/mpskit.jl_amd/ab/libmpsk_cq$v.so timeout -k 10 100 python tools/qr_only.py $shape 2>&1 | grep qrpos | sed "s/^/[$v] /" >> $O/s6_ab.log
  done
done
done
cat $O/s6_ab.log | cut -c1-110
