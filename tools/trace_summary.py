"""Summarise a rocprofv3 --kernel-trace csv: per-kernel totals inside the last `--last` fraction of the timeline, the busy
time (union of kernel intervals) and the idle gaps.  usage: trace_summary.py kernel_trace.csv [--last 0.5]"""
import csv, sys
path = sys.argv[1]
last = float(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 1.0
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t1 - (t1 - t0) * last
rows = [r for r in rows if r[0] >= lo]
span = max(r[1] for r in rows) - rows[0][0]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
agg = {}
for s, e, n in rows:
    n = n.split("(")[0][-70:]
    a = agg.setdefault(n, [0, 0])
    a[0] += 1; a[1] += e - s
print(f"window {span / 1e3:.1f} us, busy (union) {busy / 1e3:.1f} us = {100 * busy / span:.1f} %, kernels {len(rows)}")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:18]:
    print(f"  {t / 1e3:10.1f} us  {c:6d} x {t / c / 1e3:8.2f} us  {n}")
