#!/usr/bin/env python3
"""tools/pmc_one.py <rocprofv3 output dir> <kernel substring>: mean counter value per launch of the matching kernel."""
import collections, csv, glob, sys
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k}: mean {sum(v)/len(v):.1f} over {len(v)} launches")
