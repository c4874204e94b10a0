#!/usr/bin/env python3
"""Print a rocprofv3 --stats kernel_stats.csv compactly: tools/kstats.py <dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:64]:64s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
