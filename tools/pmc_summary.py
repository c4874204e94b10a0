#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection csv: per kernel, mean counter value per dispatch."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:36]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if k.startswith("__amd") : continue
            print(f"{k:36s} " + " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())) + f" (n={len(next(iter(cs.values())))})")
