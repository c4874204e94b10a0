#!/usr/bin/env python3
"""Frame time away from the poses the schedule was tuned on ("nothing protects performance away from these
poses"): scenes x camera distances x angular steps per frame, 1080p, the library's defaults, in bench.py's call order
(sort of the new pose, then its frame; nothing read back; 100 frames after 20 of warm-up).  JSON lines on stdout.

  python tools/pose_sweep.py [splats=6000000]
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

N = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
W, H = 1920, 1080
SCENES = {"clustered (cfg 2)": lambda: synth.clustered_cloud(N, seed=6), "uniform": lambda: synth.uniform_cloud(N, seed=1),
          "overlap stress (cfg 5)": lambda: synth.overlap_stress_cloud(N), "tight clusters": lambda: synth.clustered_cloud(N, seed=8, n_clusters=8, sigma=0.1, frac=0.9)}
for name, make in SCENES.items():
    splats = scene.raw_to_splats(pkg, make(), center=False)
    with pkg.Gsx(0) as g:
        g.upload(splats)
        for dist in (2.5, 5.0, 10.0):
            for step_deg in (0.0, 0.25, 3.0, 15.0):
                frames, warm = 100, 20
                poses = []
                for f in range(frames + warm):
                    world = cam.orbit_pose(dist, [((0.0, 1.0, 0.0), np.radians(step_deg))] * f) if step_deg else cam.orbit_pose(dist, [])
                    view, proj, pos = cam.camera_matrices(world, 45.0, W / H)
                    poses.append((pos, pkg.make_uniforms(proj, view)))
                    if not step_deg:
                        poses = poses * (frames + warm)
                        break
                for f in range(warm):
                    g.sort(poses[f][0], download=False); g.render(poses[f][1], W, H, download=False)
                g.sync()
                t0 = time.perf_counter()
                for f in range(warm, warm + frames):
                    g.sort(poses[f][0], download=False); g.render(poses[f][1], W, H, download=False)
                g.sync()
                dt = (time.perf_counter() - t0) / frames
                t = g.timings()
                print(json.dumps({"scene": name, "splats": N, "distance": dist, "deg_per_frame": step_deg, "ms_per_frame": dt * 1e3,
                                  "last_frame_tile_pairs": int(t.get("tile_pairs", 0)), "last_frame_iters": int(t.get("render_iters", 0))}), flush=True)
