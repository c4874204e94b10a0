#!/usr/bin/env python3
"""Merge the host's call times (GSX_TRACE_HOST=1 lines on stderr) with a rocprofv3 kernel trace: tools/host_vs_gpu.py <dir> <stderr file> [N [fraction]]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 120
ev = []
for r in csv.DictReader(open(f)):
    ev.append((int(r["Start_Timestamp"]) / 1e3, "gpu", f"q{r['Queue_Id']} dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} {r['Kernel_Name'][:40]}"))
for l in open(sys.argv[2]):
    m = re.match(r"\[host\] sort ([\d.]+) \.\. ([\d.]+)", l)
    if m:
        ev.append((float(m.group(1)), "host", f"sort enter (returns +{float(m.group(2)) - float(m.group(1)):.1f})"))
    m = re.match(r"\[host\] render ([\d.]+) wait-done ([\d.]+) \.\. ([\d.]+)", l)
    if m:
        a, b, c = map(float, m.groups())
        ev.append((a, "host", f"render enter (waited {b - a:.1f}, launches {c - b:.1f})"))
        ev.append((c, "host", "render returns"))
ev.sort()
at = int(len(ev) * (float(sys.argv[4]) if len(sys.argv) > 4 else 0.5))
t0 = ev[at][0]
for t, k, s in ev[at:at + n]:
    print(f"{t - t0:9.1f} {'    ' if k == 'gpu' else '>>> '}{s}")
