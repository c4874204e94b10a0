#!/usr/bin/env python3
"""tools/chunk_count_roofline.py <rocprofv3 --kernel-trace dir>  ->  HBM rate of the cull / projection pass (k_chunk_count).

Every launch covers a window of ranks; its grid says how many (512 threads x 2 ranks per workgroup).  Algorithmic bytes per
rank of the window: 4 B permutation entry + one 64-B scene record gathered by rank + 4 B tile box written = 72 B; a rank that
survives the cull also writes its 48-B screen-space record (at most +48 B: the upper figure).  Launches are grouped by window
size because the first (small, dense) and the later (large, mostly hidden) windows of a frame behave differently."""
import csv, glob, json, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
groups = {}
for r in csv.DictReader(open(f)):
    if not r["Kernel_Name"].startswith("k_chunk_count"):
        continue
    ranks = int(r["Grid_Size_X"]) // 512 * 1024
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    key = "<= 256 K ranks" if ranks <= 262144 else ("<= 1 M ranks" if ranks <= (1 << 20) else "> 1 M ranks")
    g = groups.setdefault(key, {"launches": 0, "ranks": 0, "us": 0.0})
    g["launches"] += 1; g["ranks"] += ranks; g["us"] += dur
out = {"kernel": "k_chunk_count (projection + cull + tile count of one iteration window)", "bytes_per_rank": "72 (+48 for a rank that survives the cull)", "windows": {}}
tot_r = tot_us = 0
for k, g in sorted(groups.items()):
    lo = 72 * g["ranks"] / (g["us"] * 1e-6) / 1e9
    out["windows"][k] = {"launches": g["launches"], "mean_ranks": g["ranks"] / g["launches"], "mean_us": g["us"] / g["launches"],
                         "hbm_gbs_at_72B": lo, "hbm_gbs_at_120B": lo * 120 / 72, "hbm_frac_at_72B": lo / 8000.0}
    tot_r += g["ranks"]; tot_us += g["us"]
if tot_us:
    out["all"] = {"hbm_gbs_at_72B": 72 * tot_r / (tot_us * 1e-6) / 1e9, "hbm_frac_at_72B": 72 * tot_r / (tot_us * 1e-6) / 1e9 / 8000.0}
print(json.dumps(out, indent=1))
