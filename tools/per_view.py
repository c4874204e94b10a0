#!/usr/bin/env python3
"""Frame time of each of the first 8 poses of bench.py's pose set (the view rank r of an N-rank run renders), one GPU."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = bench.load_pkg()
synth = importlib.import_module("gaussian_splatting_amd.synth")
scene = importlib.import_module("gaussian_splatting_amd.scene")
n, W, H = 6_000_000, 1920, 1080
splats = scene.raw_to_splats(pkg, synth.SCENES["clustered"](n), center=False)
views = bench.orbit_views(pkg, 64, W, H)
g = pkg.Gsx(0); g.upload(splats)
for v in range(8):
    u = pkg.make_uniforms(views[v][1], views[v][0])
    for rnd in range(2):
        g.sync(); t0 = time.perf_counter()
        for i in range(50):
            g.sort(views[v][2], download=False); g.render(u, W, H, download=False)
        g.sync(); dt = (time.perf_counter() - t0) / 50 * 1e3
    t = g.timings()
    print(f"view {v}: cam {views[v][2]} {dt:.3f} ms per frame, passes {t['sort_passes']}, pairs {t['tile_pairs']}, iters {t['render_iters']}", flush=True)
