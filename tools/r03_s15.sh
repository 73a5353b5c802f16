#!/bin/bash
# round 3, session 15: dominance probe (test + overhead), kernel stats of one mode-3 split
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s15
O=gpurun_out/s15
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_golden.py -x -q -m gpu -k "svd or split or bond_matrix" > $O/pytest_svd.log 2>&1 || { tail -40 $O/pytest_svd.log; exit 1; }
tail -2 $O/pytest_svd.log
for n in 4096 2048 1024; do timeout -k 10 120 python tools/svd_once.py $n graded6 4 3 2>&1 | tail -1 | tee -a $O/once.log; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/tools/svd_once.py 4096 graded6 4 3 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find $O/prof -name "*kernel_stats.csv" | head -1); do cp $f $O/kernel_stats_split_mode3.csv; done
for f in $(find $O/prof -name "*kernel_trace.csv" | head -1); do python tools/trace_summary.py $f > $O/trace_summary.txt 2>&1 || true; done
rm -rf $O/prof
head -12 $O/kernel_stats_split_mode3.csv | cut -c1-150
