import sys, importlib, ctypes as C
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
from conftest import load_pkg
import numpy as np, orc
pkg = load_pkg(); synth = importlib.import_module("gaussian_splatting_amd.synth")
O = orc.Oracle()
w, h = 240, 176
splats = O.raw_to_splat(synth.clustered_cloud(60000, seed=8, n_clusters=3, sigma=0.3, frac=0.95, centre_extent=0.5, scale_mul=4.0))
world, _ = O.orbit_world(2.5, (0.0, 1.0, 0.0), 0.35, 2)
view, proj, vp, cam = O.camera(world, 45.0, w / h)
g = pkg.Gsx(0); g.upload(splats)
perm = g.sort(cam)
u = pkg.make_uniforms(proj, view)
for rep in range(3):
    for first, growth in ((1 << 30, 0), (512, 2)):
        g.set_option("chunk_first", first); g.set_option("chunk_growth", growth)
        img = g.render(u, w, h); t = g.timings()
        print("first", first, "pairs", t["tile_pairs"], "iters", t["render_iters"], flush=True)
