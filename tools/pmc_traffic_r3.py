#!/usr/bin/env python3
"""tools/pmc_traffic_r3.py <dir made by tools/pmc_traffic.sh> [splats] -> profiles/r3_sort_traffic.json

Per-sort HBM traffic of the two-pass depth sort's kernels (k2_keygen_hist, k2_scan, k2_partition, k2_local x 3) from rocprofv3
PMC passes.  FETCH_SIZE / WRITE_SIZE are scaled by factors calibrated on a streaming copy of known size at 4 bytes per lane
(the narrowest access of these kernels; the 16-byte factor is recorded beside it)."""
import collections, csv, glob, json, os, sys

def per_kernel(d):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}

root = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6_000_000
cal_bytes = 96 << 20
groups = {"k2_keygen_hist": "k2_keygen_hist", "k2_scan": "k2_scan", "k2_partition": "k2_partition", "k2_local": "k2_local"}
alg = {"k2_keygen_hist": 16 * n, "k2_scan": None, "k2_partition": 12 * n, "k2_local": 12 * n}
res = {"splats": n, "calibration_bytes": cal_bytes, "kernels": {g: {"algorithmic_bytes": alg[g]} for g in groups}, "scale": {}}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    cal4, _ = per_kernel(os.path.join(root, "cal4_" + ctr))
    cal16, _ = per_kernel(os.path.join(root, "cal16_" + ctr))
    raw4 = next(v for k, v in cal4.items() if "k_copy4" in k)
    raw16 = next(v for k, v in cal16.items() if "k_copy16" in k)
    scale4 = cal_bytes / raw4
    res["scale"][ctr] = {"bytes_per_unit_at_4B_per_lane": scale4, "bytes_per_unit_at_16B_per_lane": cal_bytes / raw16}
    bench, cnt = per_kernel(os.path.join(root, "bench_" + ctr))
    sorts = max(cnt[k] for k in cnt if "k2_partition" in k)
    for g, pat in groups.items():
        rows = [k for k in bench if pat in k]
        raw = sum(bench[k] * cnt[k] for k in rows) / sorts          # per sort (k2_local: its three instances together)
        res["kernels"][g][ctr + "_bytes_per_sort"] = raw * scale4
    res["sorts_profiled"] = sorts
for g in groups:
    k = res["kernels"][g]
    k["hbm_bytes_per_sort"] = k["FETCH_SIZE_bytes_per_sort"] + k["WRITE_SIZE_bytes_per_sort"]
    if k["algorithmic_bytes"]:
        k["traffic_over_algorithmic"] = k["hbm_bytes_per_sort"] / k["algorithmic_bytes"]
# what bench.py's `roofline` averages over: one launch bracket = k2_partition, or the k2_local instances together
res["hbm_bytes_per_launch"] = 0.5 * (res["kernels"]["k2_partition"]["hbm_bytes_per_sort"] + res["kernels"]["k2_local"]["hbm_bytes_per_sort"])
res["algorithmic_bytes_per_launch"] = 12 * n
res["whole_sort_hbm_bytes"] = sum(res["kernels"][g]["hbm_bytes_per_sort"] for g in groups)
res["note"] = ("two-pass depth sort; FETCH_SIZE and WRITE_SIZE from separate rocprofv3 --pmc passes of `bench.py --steps 6 --warmup 2 --cpu-frames 0 "
               "--no-extras`, each scaled by the factor a 96 MiB 4-byte-per-lane streaming copy (tools/membw.hip k_copy4) gives for that counter; "
               "hbm_bytes_per_launch = mean of the partition launch and the bucket-sort launches (what bench.py's roofline brackets)")
os.makedirs("profiles", exist_ok=True)
json.dump(res, open("profiles/r3_sort_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
