#!/usr/bin/env python3
"""Experiment: how much would two frames in flight buy?  Two contexts on one GPU (each with its own copy of the scene) take
alternate frames of the cfg 2 workload; compared with one context doing all of them.  Not a product path."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch  # noqa: F401  (HIP runtime order)
pkg = bench.load_pkg()
synth = importlib.import_module("gaussian_splatting_amd.synth")
scene = importlib.import_module("gaussian_splatting_amd.scene")
W, H, n = 1920, 1080, 6_000_000
splats = scene.raw_to_splats(pkg, synth.SCENES["clustered"](n), center=False)
views = bench.orbit_views(pkg, 64, W, H)
v = views[0]
u = pkg.make_uniforms(v[1], v[0])
for nctx in (1, 2, 3):
    gs = [pkg.Gsx(0) for _ in range(nctx)]
    for g in gs:
        g.upload(splats)
    def step(i):
        g = gs[i % nctx]
        g.sort(v[2], download=False)
        g.render(u, W, H, download=False)
    for i in range(12):
        step(i)
    for g in gs:
        g.sync()
    K = 120
    t0 = time.perf_counter()
    for i in range(K):
        step(i)
    for g in gs:
        g.sync()
    dt = (time.perf_counter() - t0) / K
    print(f"contexts {nctx}: {dt*1e3:.4f} ms per frame, {n/dt/1e9:.2f} G splats/s", flush=True)
    for g in gs:
        g.close()
