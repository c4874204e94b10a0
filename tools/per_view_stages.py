#!/usr/bin/env python3
"""Stage breakdown (HIP events inside libgsx, profiling level 1) of the first 8 poses of bench.py's pose set, cfg-2 scene."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = bench.load_pkg()
synth = importlib.import_module("gaussian_splatting_amd.synth")
scene = importlib.import_module("gaussian_splatting_amd.scene")
n, W, H = 6_000_000, 1920, 1080
splats = scene.raw_to_splats(pkg, synth.SCENES["clustered"](n), center=False)
views = bench.orbit_views(pkg, 64, W, H)
g = pkg.Gsx(0); g.upload(splats); g.set_profiling(1)
for v in range(8):
    u = pkg.make_uniforms(views[v][1], views[v][0])
    acc = {}
    for i in range(12):
        g.sort(views[v][2], download=False); g.render(u, W, H, download=False)
        t = g.timings()
        if i >= 4:
            for k, x in t.items(): acc[k] = acc.get(k, 0.0) + x / 8
    print(f"view {v}: " + " ".join(f"{k}={x:.3f}" if k.endswith("_ms") else f"{k}={x:.0f}" for k, x in acc.items()), flush=True)
