#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_stats.csv: per-step time of the top kernels.  usage: prof_summary.py <dir> <steps>"""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 22
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:top]:
    print(f"{r['Name'][:86]:86s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/steps/1e3:9.1f}us/step {float(r['AverageNs'])/1e3:8.1f}us {float(r['TotalDurationNs'])/tot*100:5.1f}%")
print(f"{tot/steps/1e6:.3f} ms/step GPU busy")
