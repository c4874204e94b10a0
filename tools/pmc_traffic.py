#!/usr/bin/env python3
"""tools/pmc_traffic.py <dir made by tools/pmc_traffic.sh> [splats] -> profiles/r2_scatter_traffic.json

Per-launch HBM traffic of k_radix_scatter<u32> from rocprofv3 PMC passes.  FETCH_SIZE / WRITE_SIZE are scaled by
factors calibrated on a streaming copy with the kernel's own access width (4 bytes per lane) and a known byte count."""
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
res = {"splats": n, "calibration_bytes": cal_bytes, "counters": {}}
total = 0.0
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    cal4, _ = per_kernel(os.path.join(root, "cal4_" + ctr))
    cal16, _ = per_kernel(os.path.join(root, "cal16_" + ctr))
    raw4 = next(v for k, v in cal4.items() if "k_copy4" in k)
    raw16 = next(v for k, v in cal16.items() if "k_copy16" in k)
    bench, cnt = per_kernel(os.path.join(root, "bench_" + ctr))
    rows = {k: v for k, v in bench.items() if "k_radix_scatter<unsigned int" in k}
    calls = sum(cnt[k] for k in rows)
    raw = sum(rows[k] * cnt[k] for k in rows) / calls
    scale4 = cal_bytes / raw4                      # bytes per counter unit at 4 B per lane
    res["counters"][ctr] = {"raw_per_launch": raw, "launches": calls, "bytes_per_unit_at_4B_per_lane": scale4,
                            "bytes_per_unit_at_16B_per_lane": cal_bytes / raw16, "bytes_per_launch": raw * scale4}
    total += raw * scale4
res["hbm_bytes_per_launch"] = total
res["algorithmic_bytes_per_launch"] = (12 + 16 + 16) / 3 * n      # pass 0 generates the index (12 B per key), passes 1, 2 move key + index (16 B)
res["note"] = ("k_radix_scatter<u32>, mean over the three passes of the depth sort; FETCH_SIZE and WRITE_SIZE from separate "
               "rocprofv3 --pmc passes of `bench.py --steps 6 --warmup 2 --cpu-frames 0 --no-extras`, each scaled by the factor a 96 MiB 4-byte-per-lane "
               "streaming copy (tools/membw.hip k_copy4) gives for that counter")
os.makedirs("profiles", exist_ok=True)
json.dump(res, open("profiles/r2_scatter_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
