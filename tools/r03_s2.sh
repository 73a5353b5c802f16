#!/bin/bash
# round-3 GPU session 2: why do the Jacobi chains not overlap?  host enqueue time per sweep + kernel traces
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
for nc in 1 4; do
  echo "== chains $nc (debug timing)" >> $O/s2_dbg.log
  MPSK_SVD_CHAINS=$nc MPSK_SVD_DEBUG=1 timeout -k 10 200 python tools/svd_once.py 4096 graded6 >> $O/s2_dbg.log 2>&1
done
cd /tmp && export TMPDIR=/tmp
for nc in 1 4; do
  rm -rf $O/tr$nc
  MPSK_SVD_CHAINS=$nc timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr$nc -- python3 $R/tools/svd_once.py 4096 graded6 > $O/s2_tr$nc.log 2>&1
  f=$(find $O/tr$nc -name "*kernel_trace.csv" | head -1)
  echo "== chains $nc trace $f" >> $O/s2_trace.log
  python3 $R/tools/overlap_report.py $f --last 0.45 >> $O/s2_trace.log 2>&1
  python3 $R/tools/trace_summary.py $f --last 0.45 >> $O/s2_trace.log 2>&1
  # keep only a slice of the csv (the merge-back limit is 64 MiB)
  python3 - "$f" "$O/s2_slice$nc.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
sl = rows[int(n * 0.8): int(n * 0.8) + 1500]
with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    w.writerow(["start_us", "dur_us", "queue", "grid", "name"])
    t0 = int(sl[0]["Start_Timestamp"])
    for r in sl:
        w.writerow([round((int(r["Start_Timestamp"]) - t0) / 1e3, 2), round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 2),
                    r.get("Queue_Id", "?"), r.get("Grid_Size", r.get("Grid_Size_X", "?")), r["Kernel_Name"].split("(")[0][-40:]])
PY
  rm -rf $O/tr$nc
done
cd $R
echo "== bench to-tolerance with thick restart" > $O/s2_bench.log
( time timeout -k 10 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --early-sweeps 0 ) >> $O/s2_bench.log 2>&1
cat $O/s2_dbg.log | grep -v amdgpu.ids | tail -30; cat $O/s2_trace.log
