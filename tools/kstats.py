#!/usr/bin/env python3
"""Per-kernel average of a rocprofv3 --kernel-trace --stats csv: kernels with at least N calls.   python3 tools/kstats.py <kernel_stats.csv> [min_calls]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
mn = int(sys.argv[2]) if len(sys.argv) > 2 else 500
tot = 0.0
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    if int(r["Calls"]) >= mn:
        tot += float(r["AverageNs"]) if int(r["Calls"]) >= 900 else 0.0
        print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} x {float(r['AverageNs']) / 1000:8.2f} us")
print(f"sum of the averages of kernels with >= 900 calls: {tot / 1000:.1f} us")
