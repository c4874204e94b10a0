#!/usr/bin/env python3
"""Print the last N kernel launches of a rocprofv3 --kernel-trace run in start order with duration and
the idle gap before each: tools/ktrace.py <dir> [N [start_fraction]]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
if len(sys.argv) > 3:                      # start at this fraction of the trace instead of taking the tail
    at = int(len(rows) * float(sys.argv[3]))
    rows = rows[at:at + n]
else:
    rows = rows[-n:]
prev = None
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  end {(e - t0) / 1e3:9.1f}  q{r.get('Queue_Id', '?'):>2s}  dur {(e - s) / 1e3:8.1f}  gap {gap:6.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>9s}  {r['Kernel_Name'][:70]}")
    prev = e
