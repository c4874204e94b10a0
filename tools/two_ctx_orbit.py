#!/usr/bin/env python3
"""Experiment (round 3): would more frames in flight help the ORBIT (a new pose every frame)?  1 / 2 / 3 contexts on one GPU,
each with its own copy of the scene and two frame slots, take alternate frames of the 120-frame orbit.  Not a product path."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch  # noqa: F401  (HIP runtime order)
pkg = bench.load_pkg()
synth = importlib.import_module("gaussian_splatting_amd.synth")
scene = importlib.import_module("gaussian_splatting_amd.scene")
W, H, n = 1920, 1080, 6_000_000
splats = scene.raw_to_splats(pkg, synth.SCENES["clustered"](n), center=False)
path = bench.orbit_path(pkg, 120, W, H)
unis = [pkg.make_uniforms(p[1], p[0]) for p in path]
for nctx in (1, 2, 3):
    gs = [pkg.Gsx(0) for _ in range(nctx)]
    for g in gs:
        g.upload(splats)
    def step(i):
        g = gs[i % nctx]
        k = i % len(path)
        g.sort(path[k][2], download=False)
        g.render(unis[k], W, H, download=False)
    for i in range(12):
        step(i)
    for g in gs:
        g.sync()
    K = 120
    t0 = time.perf_counter()
    for i in range(12, 12 + K):
        step(i)
    for g in gs:
        g.sync()
    dt = (time.perf_counter() - t0) / K
    print(f"contexts {nctx}: {dt*1e3:.4f} ms per frame, {n/dt/1e9:.2f} G splats/s", flush=True)
    for g in gs:
        g.close()
