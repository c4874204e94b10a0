#!/usr/bin/env python3
"""One-GPU measurement of every BASELINE.json config shape, written as JSON lines to stdout.

  python tools/configs_table.py [cfg1 cfg3 cfg4 cfg5]      (cfg 2 is bench.py itself)

cfg 1  1 M uniform cloud, default orbit pose, depth sort only (keys + radix sort, permutation left in HBM)
cfg 3  6 M clustered scene, octree (max_depth 10, 1 splat/leaf) built on the host, 120-frame orbit about +Y:
       device LOD cut (min_screen_area 0.2 -> threshold 0.002) -> sort -> 1080p frame, nothing read back
cfg 4  20 M clustered scene, the 64-view pose set rendered by ONE GPU (what rank 0 of an N-rank run does for its share)
cfg 5  6 M overlap-stress scene at 3840x2160: whole frame, then the same frame as 8 bands of 270 rows one after another
Times are host wall clock around queued frames closed by gsx_sync (the library never returns to the host inside a frame).
"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pkg = bench.load_pkg()
synth = importlib.import_module("gaussian_splatting_amd.synth")
scene = importlib.import_module("gaussian_splatting_amd.scene")
cam = importlib.import_module("gaussian_splatting_amd.camera")
lod = importlib.import_module("gaussian_splatting_amd.lod")


def emit(rec):
    print(json.dumps(rec), flush=True)


def cfg1():
    n = 1_000_000
    splats = scene.raw_to_splats(pkg, synth.uniform_cloud(n, seed=1), center=False)
    view, proj, pos = cam.camera_matrices(cam.orbit_pose(5.0, []), 45.0, 16.0 / 9.0)
    with pkg.Gsx(0) as g:
        g.upload(splats)
        for _ in range(20):
            g.sort(pos, download=False)
        g.sync()
        k = 500
        t0 = time.perf_counter()
        for _ in range(k):
            g.sort(pos, download=False)
        g.sync()
        dt = (time.perf_counter() - t0) / k
        t = g.timings()
    emit({"config": "cfg1", "splats": n, "what": "keys + radix sort, back-to-back", "ms_per_sort": dt * 1e3, "splats_per_s": n / dt,
          "sort_passes": t["sort_passes"]})


def cfg3():
    n, W, H, frames = 6_000_000, 1920, 1080, 120
    timed = int(os.environ.get("GSX_CFG3_FRAMES", frames))      # (PMC passes: fewer frames, same orbit)
    t0 = time.perf_counter()
    tree = lod.build_octree(synth.clustered_cloud(n, seed=6))
    build_s = time.perf_counter() - t0
    poses = []
    for f in range(frames):
        world = cam.orbit_pose(5.0, [((0.0, 1.0, 0.0), 2.0 * np.pi / frames)] * f)
        view, proj, pos, vp = cam.camera_matrices(world, 45.0, W / H, want_viewproj=True)
        poses.append((pkg.make_uniforms(proj, view), pos, vp))
    with pkg.Gsx(0) as g:
        tree.upload(g)
        for o in os.environ.get("GSX_OPTS", "").split():      # A/B runs: GSX_OPTS="lod_walk=1"
            k, val = o.split("=")
            g.set_option(k, int(val))
        ms = []
        for f in range(0, frames, 30):          # cut sizes along the orbit (read back, outside the timing)
            ms.append(int(g.select_octree(poses[f][2], 0.002)))
        for f in range(10):
            g.select_octree(poses[f][2], 0.002, want_count=False); g.sort(poses[f][1], download=False); g.render(poses[f][0], W, H, download=False)
        g.sync()
        ahead = os.environ.get("GSX_AHEAD", "") != ""         # GSX_AHEAD=1: cut + sort of frame f+1 queued before frame f (option sort_ahead); measured: 0.401 against 0.395 ms here
        if ahead:
            g.set_option("sort_ahead", 1)
            g.select_octree(poses[0][2], 0.002, want_count=False)
            g.sort(poses[0][1], download=False)
        t0 = time.perf_counter()
        for f in range(timed):
            u, pos, vp = poses[f]
            if ahead:
                _, npos, nvp = poses[(f + 1) % frames]
                g.select_octree(nvp, 0.002, want_count=False)
                g.sort(npos, download=False)
            else:
                g.select_octree(vp, 0.002, want_count=False)
                g.sort(pos, download=False)
            g.render(u, W, H, download=False)
        g.sync()
        dt = (time.perf_counter() - t0) / timed
        t = g.timings()
    emit({"config": "cfg3", "splats": n, "octree_nodes": int(tree.n), "host_octree_build_s": build_s, "frames": timed,
          "what": "octree cut (threshold 0.002) -> sort -> 1080p frame per orbit step", "cut_sizes_at_frames_0_30_60_90": ms,
          "ms_per_frame": dt * 1e3, "fps": 1.0 / dt, "last_frame_tile_pairs": t["tile_pairs"], "last_frame_iters": t["render_iters"],
          "call_order": "cut + sort f+1, render f (sort_ahead)" if ahead else "cut, sort, render f",
          "opts": os.environ.get("GSX_OPTS", "")})
    tree.close()


def cfg4():
    n, W, H = 20_000_000, 1920, 1080
    splats = scene.raw_to_splats(pkg, synth.clustered_cloud(n, seed=20), center=False)
    views = bench.orbit_views(pkg, 64, W, H)
    us = [pkg.make_uniforms(v[1], v[0]) for v in views]
    with pkg.Gsx(0) as g:
        g.upload(splats)
        for o in os.environ.get("GSX_OPTS", "").split():      # A/B runs: GSX_OPTS="no_history=1 chunk_first=262144"
            k, val = o.split("=")
            g.set_option(k, int(val))
        ahead = os.environ.get("GSX_AHEAD", "") != ""
        if ahead:
            g.set_option("sort_ahead", 1)
        for rnd in range(2):
            if ahead:
                g.set_option("sort_ahead", 1)                  # (forgets what waits from the round before)
                g.sort(views[0][2], download=False)
            t0 = time.perf_counter()
            for i, v in enumerate(views):
                g.sort(views[(i + 1) % len(views)][2] if ahead else v[2], download=False)
                g.render(us[i], W, H, download=False)
            g.sync()
            dt = (time.perf_counter() - t0) / len(views)
        t = g.timings()
    emit({"config": "cfg4 (one GPU's work for all 64 views)", "splats": n, "views": len(views), "ms_per_view": dt * 1e3,
          "views_per_s": 1.0 / dt, "splats_per_s": n / dt, "sort_passes": t["sort_passes"], "opts": os.environ.get("GSX_OPTS", "")})


def cfg5():
    n, W, H = 6_000_000, 3840, 2160
    splats = scene.raw_to_splats(pkg, synth.overlap_stress_cloud(n, seed=4), center=False)
    view, proj, pos = cam.camera_matrices(cam.orbit_pose(5.0, []), 45.0, W / H)
    u = pkg.make_uniforms(proj, view)
    bands = [(y, min(y + 270, H)) for y in range(0, H, 270)]
    with pkg.Gsx(0) as g:
        g.upload(splats)
        out = {}
        for name, parts in (("whole_frame", [(0, H)]), ("8_bands_sequential", bands)):
            for rnd in range(3):
                g.sync()
                t0 = time.perf_counter()
                k = 10
                for _ in range(k):
                    g.sort(pos, download=False)
                    for (y0, y1) in parts:
                        g.render(u, W, H, y0=y0, y1=y1, download=False)
                g.sync()
                dt = (time.perf_counter() - t0) / k
            out[name + "_ms"] = dt * 1e3
            out[name + "_tile_pairs_last_render"] = g.timings()["tile_pairs"]
        # what ONE of 8 ranks does per frame (N = 8 rehearsed band by band on one GPU): sort + its band, with the scene sorted
        # whole (replicated sort) or culled to the band first (gsx_select_band); the slowest band sets the frame time at N = 8
        for name, cull in (("per_rank_replicated_sort", False), ("per_rank_band_cull", True)):
            per = []
            for (y0, y1) in bands:
                for rnd in range(2):
                    g.sync()
                    t0 = time.perf_counter()
                    k = 20
                    for _ in range(k):
                        if cull:
                            g.select_band(u, W, H, y0, y1, want_count=False)
                        g.sort(pos, download=False)
                        g.render(u, W, H, y0=y0, y1=y1, download=False)
                    g.sync()
                    dt = (time.perf_counter() - t0) / k
                per.append(dt * 1e3)
                if cull:
                    g.select_all()
            out[name + "_ms_per_band"] = [round(x, 4) for x in per]
            out[name + "_slowest_band_ms"] = max(per)
    emit({"config": "cfg5 (one GPU)", "splats": n, "width": W, "height": H, **out})


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg1", "cfg3", "cfg4", "cfg5"]
    for name in which:
        {"cfg1": cfg1, "cfg3": cfg3, "cfg4": cfg4, "cfg5": cfg5}[name]()
