"""Idle time between consecutive kernels of each HIP queue in the tail of a bench kernel trace (development aid).
usage: python tools/gap_analysis.py kernel_trace.csv [n_last_kernels]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n_last:]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
byq = defaultdict(list)
for r in rows:
    byq[r.get("Queue_Id", "0")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
print(f"window {1e-6 * (t1 - t0):.3f} ms, {len(rows)} kernels, {len(byq)} queues")
# union of busy intervals over all queues
iv = sorted((a, b) for q in byq.values() for a, b, _ in q)
busy, cur_a, cur_b = 0, iv[0][0], iv[0][1]
for a, b in iv[1:]:
    if a > cur_b:
        busy += cur_b - cur_a
        cur_a, cur_b = a, b
    else:
        cur_b = max(cur_b, b)
busy += cur_b - cur_a
print(f"device busy (any queue) {100.0 * busy / (t1 - t0):.1f} % of the window")
for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    ks.sort()
    run = sum(b - a for a, b, _ in ks)
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    pos = [g for g in gaps if g > 0]
    print(f"queue {q}: {len(ks)} kernels, running {100.0 * run / (t1 - t0):.1f} %, "
          f"median gap {sorted(pos)[len(pos) // 2] / 1e3 if pos else 0:.2f} us, gaps < 20 us sum "
          f"{sum(g for g in pos if g < 20000) / 1e3 / max(len(ks), 1):.2f} us per kernel")
