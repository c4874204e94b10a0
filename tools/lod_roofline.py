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
looked, emitted = [], []
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
cuts = next(c for n, (c, a) in stats.items() if n.startswith("k_compact_onepass") or n.startswith("k_compact_write"))
walk_us = sum(total_us(k, cuts) for k in ("k_octree_top", "k_octree_level", "k_octree_head", "k_octree_tail"))
comp_us = sum(total_us(k, cuts) for k in ("k_compact_count", "k_scan_block_sums", "k_compact_write", "k_compact_onepass"))
names = sorted(n.split("(")[0] for n in stats if any(n.startswith(k) or (" " + k) in n for k in ("k_octree", "k_compact", "k_scan_block")))
L, E, N = float(np.mean(looked)), float(np.mean(emitted)), int(tree.n)
walk_bytes = L * (4 + 33 + 4) + (L - 1) * 4 + E
comp_bytes = 2 * N + 4 * E
print(json.dumps({"tree_nodes": N, "nodes_looked_at": looked, "nodes_emitted": emitted,
                  "launches": names, "walk": {"kernels": "k_octree_head + k_octree_tail (or k_octree_top + k_octree_level x levels)", "us_per_cut": walk_us, "bytes_per_cut": walk_bytes,
                           "hbm_gbs": walk_bytes / (walk_us * 1e-6) / 1e9, "hbm_frac": walk_bytes / (walk_us * 1e-6) / 1e9 / 8000.0,
                           "bound": "launch + latency (one dependent launch per level) and VALU (~430 instructions per stop test), not HBM"},
                  "compaction": {"kernels": "k_compact_onepass (or k_compact_count + k_scan_block_sums + k_compact_write)", "us_per_cut": comp_us, "bytes_per_cut": comp_bytes,
                                 "hbm_gbs": comp_bytes / (comp_us * 1e-6) / 1e9, "hbm_frac": comp_bytes / (comp_us * 1e-6) / 1e9 / 8000.0,
                                 "bound": "launch latency (three dependent launches over 8.5 MB of flags)"},
                  "all_nodes_tested_instead": {"bytes": N * 34.0, "note": "what round 1's k_octree_stop read per cut"}}, indent=1))
tree.close()
