"""Debug: small render vs oracle, where do errors sit."""
import sys, importlib
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
from conftest import load_pkg
import numpy as np, orc
pkg = load_pkg(); synth = importlib.import_module("gaussian_splatting_amd.synth")
O = orc.Oracle()
w, h, n = 256, 192, 3000
splats = O.raw_to_splat(synth.uniform_cloud(n, seed=n % 13 + 1, half_extent=1.5))
world, _ = O.orbit_world(5.0, (0.0, 1.0, 0.0), 0.35, 0)
view, proj, vp, cam = O.camera(world, 45.0, w / h)
g = pkg.Gsx(0); g.upload(splats)
perm0 = None
for trial in range(5):
    if trial < 3:
        perm = g.sort(cam)
    if perm0 is None: perm0 = perm.copy()
    print("perm same", np.array_equal(perm, perm0))
    img = g.render(pkg.make_uniforms(proj, view), w, h)
    if trial == 0:
        ref = O.render(splats, perm, orc.make_uniforms(proj, view), w, h)
    err = np.abs(img.astype(np.int32) - ref.astype(np.int32)).max(axis=2)
    t = g.timings()
    bad = err > 2
    ty, tx = np.nonzero(bad.reshape(h // 16, 16, w // 16, 16).any(axis=(1, 3)))
    print("trial", trial, "max err", err.max(), "bad px", bad.sum(), "pairs", t["tile_pairs"], "iters", t["render_iters"], "bad tiles", list(zip(ty.tolist(), tx.tolist()))[:20], flush=True)
