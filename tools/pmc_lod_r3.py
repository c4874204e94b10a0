#!/usr/bin/env python3
"""tools/pmc_lod_r3.py <dir made by tools/pmc_lod_r3.sh> -> profiles/r3_lod_traffic.json

HBM traffic per cut of the octree cut's three launches at cfg 3 (8.48 M nodes) from rocprofv3 PMC passes, FETCH_SIZE / WRITE_SIZE
scaled by factors calibrated on streaming copies of known size (16 bytes per lane for the walk's 2 x 16-byte node records and the
compaction's 16-byte flag loads; the 4-byte factor beside it), against the algorithmic bytes of profiles/r3_cfg3_lod_roofline.json."""
import collections, csv, glob, json, os, sys

def per_kernel(d):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}

root = sys.argv[1]
cal_bytes = 96 << 20
alg = json.load(open("profiles/r3_cfg3_lod_roofline.json"))
groups = {"k_octree_head": "k_octree_head", "k_octree_tail": "k_octree_tail", "k_compact_onepass": "k_compact_onepass"}
res = {"tree_nodes": alg["tree_nodes"], "kernels": {g: {} for g in groups}, "scale": {}}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    cal4, _ = per_kernel(os.path.join(root, "cal4_" + ctr))
    cal16, _ = per_kernel(os.path.join(root, "cal16_" + ctr))
    s4 = cal_bytes / next(v for k, v in cal4.items() if "k_copy4" in k)
    s16 = cal_bytes / next(v for k, v in cal16.items() if "k_copy16" in k)
    res["scale"][ctr] = {"bytes_per_unit_at_4B_per_lane": s4, "bytes_per_unit_at_16B_per_lane": s16}
    vals, cnt = per_kernel(os.path.join(root, "cfg3_" + ctr))
    for g, pat in groups.items():
        rows = [k for k in vals if pat in k]
        res["kernels"][g][ctr + "_bytes_per_cut"] = sum(vals[k] for k in rows) * s16
        res["kernels"][g]["launches_profiled"] = sum(cnt[k] for k in rows)
for g in groups:
    k = res["kernels"][g]
    k["hbm_bytes_per_cut"] = k["FETCH_SIZE_bytes_per_cut"] + k["WRITE_SIZE_bytes_per_cut"]
walk = res["kernels"]["k_octree_head"]["hbm_bytes_per_cut"] + res["kernels"]["k_octree_tail"]["hbm_bytes_per_cut"]
res["walk"] = {"hbm_bytes_per_cut": walk, "algorithmic_bytes_per_cut": alg["walk"]["bytes_per_cut"], "traffic_over_algorithmic": walk / alg["walk"]["bytes_per_cut"]}
comp = res["kernels"]["k_compact_onepass"]["hbm_bytes_per_cut"]
res["compaction"] = {"hbm_bytes_per_cut": comp, "algorithmic_bytes_per_cut": alg["compaction"]["bytes_per_cut"], "traffic_over_algorithmic": comp / alg["compaction"]["bytes_per_cut"]}
res["note"] = ("cfg 3, the first 24 frames of the orbit, one stream, one frame at a time; FETCH_SIZE and WRITE_SIZE from separate rocprofv3 --pmc passes "
               "of tools/configs_table.py cfg3, scaled by the factor a 96 MiB 16-byte-per-lane streaming copy (tools/membw.hip k_copy16) gives for that counter")
json.dump(res, open("profiles/r3_lod_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
