#!/bin/bash
# round-3 GPU session 4: where does the time go at small bond dimension (configs 2 and 3)?
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for cfg in c2 c3; do
  rm -rf $O/p_$cfg
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$cfg -- python3 $R/tools/bench_configs.py $cfg > $O/s4_$cfg.log 2>&1
  f=$(find $O/p_$cfg -name "*kernel_stats.csv" | head -1)
  echo "== $cfg" >> $O/s4_stats.log
  grep -v amdgpu.ids $O/s4_$cfg.log | tail -3 >> $O/s4_stats.log
  head -24 $f | cut -c1-220 >> $O/s4_stats.log
  t=$(find $O/p_$cfg -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/trace_summary.py $t --last 0.4 >> $O/s4_stats.log 2>&1
  rm -rf $O/p_$cfg
done
cd $R
python tools/bench_dac.py > $O/s4_dac.log 2>&1
cat $O/s4_stats.log; grep -v amdgpu $O/s4_dac.log | tail -20
