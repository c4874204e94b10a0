#!/usr/bin/env python3
"""tools/lod_roofline.py <kernel_stats dir of tools/lod_profile.sh>  ->  bytes the octree cut of cfg 3 moves, against its kernel time.

Counts, on the host, the nodes the top-down cut LOOKS AT (the emitted nodes and all their ancestors) for four poses of the
orbit, prices them (4 B queue entry + 33 B box and leaf flag + 4 B child word read, 4 B queue entry written per non-root node,
1 B emit stamp per emitted node; compaction: 1 B flag per node of the tree read twice, 4 B id per emitted node written) and
divides by the mean kernel durations of the rocprofv3 summary."""
import csv, glob, importlib, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = bench.load_pkg()
synth = importlib.import_module("gaussian_splatting_amd.synth")
cam = importlib.import_module("gaussian_splatting_amd.camera")
lod = importlib.import_module("gaussian_splatting_amd.lod")
stats = {}
for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0])):
    stats[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
def total_us(prefix, per_frame_calls):
    return sum(avg * calls for name, (calls, avg) in stats.items() if name.startswith(prefix) or (" " + prefix) in name) / per_frame_calls
tree = lod.build_octree(synth.clustered_cloud(6_000_000, seed=6))
# level starts (BFS numbering: a level ends where the parent ids pass its own start)
LV = [0, 1]
while LV[-1] < tree.n:
    nxt = int(np.searchsorted(tree.parent[1:].astype(np.int64), LV[-1])) + 1
    if nxt <= LV[-1]: break
    LV.append(nxt)
looked, emitted, queue_len = [], [], []
with pkg.Gsx(0) as g:
    tree.upload(g)
    for f in (0, 30, 60, 90):
        world = cam.orbit_pose(5.0, [((0.0, 1.0, 0.0), 2.0 * np.pi / 120)] * f)
        view, proj, pos, vp = cam.camera_matrices(world, 45.0, 1920 / 1080, want_viewproj=True)
        m = g.select_octree(vp, 0.002)
        cut = g.selection()
        seen = np.zeros(tree.n, bool)
        cur = cut
        while len(cur):
            cur = cur[~seen[cur]]
            seen[cur] = True
            cur = np.unique(tree.parent[cur])
            cur = cur[cur != 0xFFFFFFFF]
        looked.append(int(seen.sum())); emitted.append(int(m))
        lv7 = (np.arange(tree.n) >= LV[7]) & (np.arange(tree.n) < LV[8]) if len(LV) > 8 else np.zeros(tree.n, bool)
        queue_len.append(int((seen & lv7).sum()))
cuts = next(c for n, (c, a) in stats.items() if n.startswith("k_compact_onepass") or n.startswith("k_compact_write"))
walk_us = sum(total_us(k, cuts) for k in ("k_octree_top", "k_octree_level", "k_octree_head", "k_octree_tail"))
comp_us = sum(total_us(k, cuts) for k in ("k_compact_count", "k_scan_block_sums", "k_compact_write", "k_compact_onepass"))
names = sorted(n.split("(")[0] for n in stats if any(n.startswith(k) or (" " + k) in n for k in ("k_octree", "k_compact", "k_scan_block")))
L, E, N = float(np.mean(looked)), float(np.mean(emitted)), int(tree.n)
three = any(n.startswith("k_octree_head") for n in stats)
if three:   # 32-byte node record per looked-at node, the one global queue (level 7 of cfg 3: written and read once), 1 B stamp per emitted node
    q = float(np.mean(queue_len))
    walk_bytes = L * 32 + q * 8 + E
    comp_bytes = N + 4 * E                  # flags read once, ids written
else:
    walk_bytes = L * (4 + 33 + 4) + (L - 1) * 4 + E
    comp_bytes = 2 * N + 4 * E
# head start -> end of the compaction, per cut, from the kernel trace of the same run (includes the gaps between the launches)
spans = []
tr = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if tr:
    rows = sorted(csv.DictReader(open(tr[0])), key=lambda r: int(r["Start_Timestamp"]))
    first = "k_octree_head" if three else "k_octree_top"
    last = "k_compact_onepass" if three else "k_compact_write"
    cur = None
    for r in rows:
        if r["Kernel_Name"].startswith(first): cur = int(r["Start_Timestamp"])
        elif cur is not None and r["Kernel_Name"].startswith(last):
            spans.append((int(r["End_Timestamp"]) - cur) / 1e3); cur = None
print(json.dumps({"tree_nodes": N, "nodes_looked_at": looked, "nodes_emitted": emitted, "launches_per_cut": names,
                  "cut_us_first_launch_start_to_last_end": {"median": float(np.median(spans)) if spans else None, "mean": float(np.mean(spans)) if spans else None, "cuts": len(spans)},
                  "walk": {"kernels": "k_octree_head + k_octree_tail" if three else "k_octree_top + k_octree_level x levels", "us_per_cut": walk_us, "bytes_per_cut": walk_bytes,
                           "hbm_gbs": walk_bytes / (walk_us * 1e-6) / 1e9, "hbm_frac": walk_bytes / (walk_us * 1e-6) / 1e9 / 8000.0,
                           "bound": "latency of dependent rounds (a node's record must arrive before its children are known) and request rate of 32-byte gathers, not HBM bytes"},
                  "compaction": {"kernels": "k_compact_onepass" if three else "k_compact_count + k_scan_block_sums + k_compact_write", "us_per_cut": comp_us, "bytes_per_cut": comp_bytes,
                                 "hbm_gbs": comp_bytes / (comp_us * 1e-6) / 1e9, "hbm_frac": comp_bytes / (comp_us * 1e-6) / 1e9 / 8000.0,
                                 "bound": "latency: load, scan, ~3 us until the other workgroups' counts are visible, write (tools/compactbench.hip); the flags arrive cold from the walk's scattered byte stores"},
                  "all_nodes_tested_instead": {"bytes": N * 34.0, "note": "what round 1's k_octree_stop read per cut"}}, indent=1))
tree.close()
