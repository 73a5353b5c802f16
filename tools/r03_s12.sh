#!/bin/bash
# round 3, session 12: config-4 sweep with the truncation-aware split; eig2 intra-step skipping under mode 3; kernel stats of one split
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s12
O=gpurun_out/s12
timeout -k 10 400 python tools/bench_configs.py c4sweep:64:1024:2 2>&1 | tee $O/c4sweep.log
for r in 0 1e-2 1; do
  for n in 4096 2048 1024; do
    MPSK_SVD_INTRA=$r timeout -k 10 120 python tools/svd_once.py $n graded6 3 3 2>&1 | tail -1 | sed "s/^/intra=$r /" | tee -a $O/intra_mode3.log
  done
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/tools/svd_once.py 4096 graded6 4 3 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find $O/prof -name "*kernel_stats.csv" | head -1); do cp $f $O/kernel_stats_split_mode3.csv; done
rm -rf $O/prof
head -14 $O/kernel_stats_split_mode3.csv | cut -c1-160
