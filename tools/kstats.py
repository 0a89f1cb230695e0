#!/usr/bin/env python3
"""print a rocprofv3 *_kernel_stats.csv: calls per forward, average us, share.   Usage: python tools/kstats.py file.csv [launches_divisor] [rows]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
div = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = int(sys.argv[3]) if len(sys.argv) > 3 else 24
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total {tot / div / 1e6:.3f} ms per pass")
for r in rows[:n]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']) // div:>5d} {float(r['AverageNs']) / 1e3:9.1f}us {float(r['TotalDurationNs']) / tot * 100:6.2f}%")
