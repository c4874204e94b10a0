#!/usr/bin/env python3
"""Per-frame look at the rank caps along the 120-frame orbit of cfg 2: capped?, redo continuations, pairs, iterations.
  python tools/orbit_caps.py [frames=40] [opt=value ...]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = bench.load_pkg()
synth = importlib.import_module("gaussian_splatting_amd.synth")
scene = importlib.import_module("gaussian_splatting_amd.scene")
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
W, H = 1920, 1080
splats = scene.raw_to_splats(pkg, synth.SCENES["clustered"](6_000_000), center=False)
path = bench.orbit_path(pkg, 120, W, H)
g = pkg.Gsx(0)
g.upload(splats)
for o in sys.argv[2:]:
    k, v = o.split("=")
    g.set_option(k, int(v))
rows = []
for i in range(frames):
    view, proj, pos = path[i % 120]
    g.sort(pos, download=False)
    g.render(pkg.make_uniforms(proj, view), W, H, download=False)
    if i >= 2:
        g.sync()          # (serializes: this tool looks at what each frame did, not at how fast)
        t = g.timings()
        rows.append((i, t["render_capped"], t["render_redos"], t["tile_pairs"], t["render_iters"]))
for r in rows:
    print("frame %3d capped %d redos %d pairs %9d iters %d" % r)
g.close()
