"""How much of a rocprofv3 --kernel-trace timeline has >= 2 kernels in flight (cross-stream overlap), per kernel name.
usage: overlap_report.py kernel_trace.csv [--last 0.5]"""
import csv, sys
path = sys.argv[1]
last = float(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 1.0
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:], r.get("Queue_Id", r.get("Stream_Id", "?"))))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t1 - (t1 - t0) * last
rows = [r for r in rows if r[0] >= lo]
ev = []
for s, e, n, q in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, prev, busy, over = 0, ev[0][0], 0, 0
for t, d in ev:
    if depth >= 1: busy += t - prev
    if depth >= 2: over += t - prev
    depth += d; prev = t
span = max(r[1] for r in rows) - rows[0][0]
print(f"window {span/1e3:.0f} us, busy {busy/1e3:.0f} us, >=2 kernels in flight {over/1e3:.0f} us ({100*over/max(busy,1):.1f} % of busy), sum of durations {sum(e-s for s,e,_,_ in rows)/1e3:.0f} us")
qs = {}
for s, e, n, q in rows:
    qs.setdefault(q, [0, 0]); qs[q][0] += 1; qs[q][1] += e - s
for q, (c, t) in qs.items():
    print(f"  queue {q}: {c} kernels, {t/1e3:.0f} us")
