#!/usr/bin/env python3
"""Reads a rocprofv3 kernel trace (csv) and reports how much kernels of different queues/streams overlapped in time.
  python tools/trace_overlap.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import sys
from collections import Counter

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("columns:", list(rows[0].keys()))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?"), r["Kernel_Name"][:40]) for r in rows)
tail = ev[len(ev) * 3 // 4:]                   # the last quarter: the second "concurrent" timing
print("queues:", Counter(e[2] for e in tail), "streams:", Counter(e[3] for e in tail))
busy = 0
union = 0
cur_end = tail[0][0]
for s, e, *_ in tail:
    busy += e - s
    if e > cur_end:
        union += e - max(s, cur_end)
        cur_end = e
span = tail[-1][1] - tail[0][0]
print(f"last quarter: span {span / 1e6:.2f} ms, sum of kernel durations {busy / 1e6:.2f} ms, union of busy intervals {union / 1e6:.2f} ms "
      f"-> average concurrency {busy / union:.2f}")
for s, e, q, st, name in tail[1000:1030]:
    print(f"  {(s - tail[1000][0]) / 1e3:9.2f} us +{(e - s) / 1e3:7.2f}  q{q} s{st} {name}")
