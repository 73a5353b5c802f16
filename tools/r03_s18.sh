#!/bin/bash
# round 3, session 18: why keep-half splits are slow in mode 2 and crash in mode 3 at 2048
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s18
O=gpurun_out/s18
MPSK_SPLIT_MAXFRAC=0.76 timeout -k 10 100 python tools/svd_half.py 2048 graded6 3 > $O/crash.log 2>&1
echo "rc=$?" >> $O/crash.log
tail -15 $O/crash.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/tools/svd_half.py 2048 graded6 2 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find $O/prof -name "*kernel_stats.csv" | head -1); do cp $f $O/kernel_stats_half.csv; done
rm -rf $O/prof
head -8 $O/kernel_stats_half.csv | cut -c1-170
