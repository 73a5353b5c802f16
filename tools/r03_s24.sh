#!/bin/bash
# round 3, session 24: where a complex real-time TDVP step spends its time (L = 24, D = 512)
set -e
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s24
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py ctdvp:24:512 > $O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find $O/prof -name "*kernel_stats.csv" | head -1); do cp $f $O/kernel_stats_ctdvp.csv; done
for f in $(find $O/prof -name "*kernel_trace.csv" | head -1); do python tools/trace_summary.py $f > $O/trace_summary.txt 2>&1 || true; done
rm -rf $O/prof
grep ctdvp $O/prof.log
head -24 $O/trace_summary.txt | cut -c1-140
