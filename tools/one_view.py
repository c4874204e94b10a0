#!/usr/bin/env python3
"""Render one pose of bench.py's pose set repeatedly (for rocprofv3 --kernel-trace):  python3 tools/one_view.py VIEW [FRAMES]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = bench.load_pkg()
synth = importlib.import_module("gaussian_splatting_amd.synth")
scene = importlib.import_module("gaussian_splatting_amd.scene")
v = int(sys.argv[1]); frames = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n, W, H = 6_000_000, 1920, 1080
splats = scene.raw_to_splats(pkg, synth.SCENES["clustered"](n), center=False)
views = bench.orbit_views(pkg, 64, W, H)
g = pkg.Gsx(0); g.upload(splats)
u = pkg.make_uniforms(views[v][1], views[v][0])
g.sync(); t0 = time.perf_counter()
for i in range(frames):
    g.sort(views[v][2], download=False); g.render(u, W, H, download=False)
g.sync()
print(f"view {v}: {(time.perf_counter() - t0) / frames * 1e3:.3f} ms per frame", g.timings(), flush=True)
