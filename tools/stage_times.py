#!/usr/bin/env python3
"""Per-stage device timings (HIP events inside libgsx) for one synthetic scene."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from conftest import load_pkg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=6_000_000)
ap.add_argument("--scene", default="clustered")
ap.add_argument("--w", type=int, default=1920)
ap.add_argument("--h", type=int, default=1080)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--sort-only", action="store_true")
ap.add_argument("--no-profiling", action="store_true")
ap.add_argument("--opt", action="append", default=[], help="name=value for gsx_set_option")
a = ap.parse_args()

pkg = load_pkg()
import importlib
synth = importlib.import_module("gaussian_splatting_amd.synth")
import orc
O = orc.Oracle()
t0 = time.time()
raw = synth.SCENES[a.scene](a.n)
splats = O.raw_to_splat(raw)
print(f"scene {a.scene} n={a.n} built in {time.time()-t0:.1f}s", flush=True)
world, _ = O.orbit_world(5.0)
view, proj, vp, cam = O.camera(world, 45.0, a.w / a.h)
g = pkg.Gsx(0)
g.upload(splats)
g.set_profiling(not a.no_profiling)
for o in a.opt:
    k, v = o.split('=')
    g.set_option(k, int(v))
u = pkg.make_uniforms(proj, view)
for it in range(a.iters):
    t0 = time.perf_counter()
    g.sort(cam, download=False)
    if not a.sort_only:
        g.render(u, a.w, a.h, download=False)
    g.sync()
    wall = (time.perf_counter() - t0) * 1e3
    t = g.timings()
    print(f"it{it}: wall {wall:.3f} ms | " + " ".join(f"{k}={v:.3f}" if isinstance(v, float) else f"{k}={v}" for k, v in t.items()), flush=True)
