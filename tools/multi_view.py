#!/usr/bin/env python3
"""Frame time when one context cycles through several camera views (what a rank of `bench.py --gpus N` does)."""
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
for stride in (1, 8):
    mine = views[0::stride][:8]
    u = [pkg.make_uniforms(v[1], v[0]) for v in mine]
    for rnd in range(4):
        t0 = time.perf_counter()
        for i in range(len(mine)):
            g.sort(mine[i][2], download=False); g.render(u[i], W, H, download=False); g.timings()
        dt = (time.perf_counter() - t0) / len(mine) * 1e3
        print(f"stride {stride} round {rnd}: {dt:.3f} ms per frame", flush=True)
